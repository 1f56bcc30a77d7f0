#!/usr/bin/env python3
"""Generator of the WIDE pairing programs (bls12_381_amd/csrc/wide.hip.h): one pairing on a whole workgroup.

The batched kernels give a pairing 2 or 4 lanes; one pairing then takes 6-12 ms however small the batch, because a lone
wavefront issues one multiply-add per ~10 cycles.  For small batches the arithmetic of ONE pairing is spread over the 1024 (or
512) lanes of a workgroup instead.  What runs there is not tower code but a straight-line PROGRAM of rounds over Fp values that live in
LDS slots; in a round every lane computes (part of) one

    SOP   out = (sum_t x_t * y_t) / R'      one Montgomery reduction for the whole sum (the reference's sum_of_products idea,
                                            fp.rs:430-484); x_t, y_t are small linear combinations of slots, formed while loading
    LIN   out = weak_reduce(sum_i c_i s_i)  a linear combination made a stored value
    INV   out = 1 / s                       the one inversion of the final exponentiation

with every product of an SOP dealt to a quad (or pair) of lanes -- four (eight) limbs of x each -- that add their unreduced column
sums to the SOP's accumulator in LDS, from where ONE lane reduces.  This script writes the reference's
Miller loop (pairings.rs:668-770) and final exponentiation (:48-176) over a symbolic Fp type -- Fp12 in the basis 1, w, ..., w^5
over Fp2 (w^2 = v, w^6 = u + 1), where a product coefficient is ONE sum of twelve Fp products -- levels the resulting DAG into
rounds, deals lanes, allocates slots by liveness and emits the tables the kernel interprets.  Every node carries its exact
value for a test input, so the formulas are checked against oracle/bls12_381_ref.py while the program is generated (--check),
and tests/test_wide_program.py re-runs the ENCODED tables at limb level against the oracle.

    python tools/gen_wide_prog.py [--check] [--out bls12_381_amd/wide_prog.bin] [--lanes N --chunk K]

Without --lanes / --chunk the blob holds the programs of both configurations built into the library (CONFIGS).
"""
import os
import struct
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
BLS_X = 0xd201000000010000
NL, LW = 14, 28
RP = 1 << (NL * LW)                    # the Montgomery factor of the internal form (fe.hip.h)
V_DIV = 2520                           # floor(2^392 / p)
MAX_A_SUM = 17                         # sum over the terms of an SOP of A_x * A_y (column sums stay below 2^64)
MAX_A = 15                             # limb bound of a lazily formed operand (28-bit limbs in 32-bit words)
VB = 4                                 # negation adds VB * p (spread over the limbs); every stored value is weakly reduced below 2p
COEF = [0, 1, -1, 2, -2, 3, -3, 4, -4, 6, -6, 8, -8, 12, -12, 9]
LANES = 1024                           # lanes of the workgroup (wide.hip.h WIDE_LANES)
CHUNK = 4                              # limbs of x per product lane (wide.hip.h WIDE_K): every product is dealt to ceil(14 / CHUNK) lanes
MAX_ACC = 128                          # sums of products per round (column accumulators in LDS)
MAX_SLOTS = 256
POST_CYCLOTOMIC = False
OP_NOP, OP_SOP, OP_LIN, OP_INV = 0, 1, 2, 3


class Node:
    __slots__ = ("kind", "id", "val", "V", "level", "terms", "src", "slot", "fixed_slot", "post", "rv")

    def __init__(self, kind, val, V):
        self.kind, self.val, self.V = kind, val % P, V
        self.level, self.terms, self.src, self.slot, self.fixed_slot, self.post, self.rv = 0, None, None, None, None, None, False


class Lin:
    """lazy linear combination of stored values: {node: small integer}"""
    __slots__ = ("c",)

    def __init__(self, c=None):
        self.c = {k: v for k, v in (c or {}).items() if v}

    @property
    def val(self):
        return sum(n.val * k for n, k in self.c.items()) % P

    def A(self):
        return sum(abs(k) * (2 if k < 0 else 1) for k in self.c.values())

    def V(self):
        return sum(abs(k) * (VB if k < 0 else n.V) for n, k in self.c.items())

    def level(self):
        return max([n.level for n in self.c] + [0])

    def __add__(self, o):
        d = dict(self.c)
        for n, k in o.c.items():
            d[n] = d.get(n, 0) + k
        return Lin(d)

    def scale(self, k):
        return Lin({n: v * k for n, v in self.c.items()})

    def encodable(self):
        return len(self.c) <= 2 and all(k in COEF for k in self.c.values()) and self.A() <= MAX_A


class E:
    """an Fp value under construction: a linear part plus a list of product terms (x, y), both Lin"""
    __slots__ = ("lin", "terms", "memo")

    def __init__(self, lin=None, terms=None):
        self.lin, self.terms, self.memo = lin or Lin(), terms or [], None

    @property
    def val(self):
        return (self.lin.val + sum(x.val * y.val for x, y in self.terms)) % P

    def __add__(self, o):
        o = as_e(o)
        return E(self.lin + o.lin, self.terms + o.terms)

    def __neg__(self):
        return E(self.lin.scale(-1), [(x.scale(-1), y) for x, y in self.terms])

    def __sub__(self, o):
        return self + (-as_e(o))

    def __rmul__(self, k):               # small integer times value
        return self.times(k)

    def times(self, k):
        return E(self.lin.scale(k), [(x.scale(k), y) for x, y in self.terms])

    def __mul__(self, o):
        if isinstance(o, int):
            return self.times(o)
        return E(None, [(G.as_lin(self), G.as_lin(o))])


def as_e(x):
    return x if isinstance(x, E) else E(Lin({x: 1}))


class Graph:
    def __init__(self):
        self.nodes, self.consts, self.inputs, self.outputs = [], {}, [], []
        self.one = self.const(1)

    def _add(self, n):
        n.id = len(self.nodes)
        self.nodes.append(n)
        return n

    def inp(self, val, slot):
        n = self._add(Node("in", val, 1))
        n.fixed_slot = slot
        self.inputs.append(n)
        return as_e(n)

    def const(self, val):
        val %= P
        if val not in self.consts:
            self.consts[val] = self._add(Node("const", val, 1))
        return self.consts[val]

    def cst(self, val):
        return as_e(self.const(val))

    # ---- materialisation ---------------------------------------------------------------------------------------------
    def lin_node(self, lin):
        """LIN node(s): a linear combination becomes ONE stored value below 2p.  Coefficients outside the table are split,
        long combinations are folded four slots at a time."""
        if len(lin.c) == 1:
            (n, k), = lin.c.items()
            if k == 1:
                return n
        items = []
        for n, k in lin.c.items():
            sgn, mag = (1 if k > 0 else -1), abs(k)
            while mag:
                big = max(c for c in COEF if 0 < c <= mag and sgn * c in COEF and c * (2 if sgn < 0 else 1) <= MAX_A - 1)   # room for the folded head
                items.append((n, sgn * big))
                mag -= big
        while True:
            head, a = [], 0
            while items and len(head) < 4:
                n, k = items[0]
                w = abs(k) * (2 if k < 0 else 1)
                if a + w > MAX_A:
                    break
                head.append(items.pop(0)); a += w
            assert head
            m = self._add(Node("lin", sum(n.val * k for n, k in head), 2))
            m.src = head
            m.level = max(n.level for n, _ in head) + 1
            assert sum(abs(k) * (VB if k < 0 else n.V) for n, k in head) <= 1024
            if not items:
                return m
            items.insert(0, (m, 1))

    def as_lin(self, e):
        """an operand of a product: a Lin of at most two stored values"""
        e = as_e(e)
        if e.memo is None:                                 # an unreduced value used as an operand several times is reduced once
            lin = self.reduce(e) if e.terms else e.lin
            if not lin.encodable():
                lin = Lin({self.lin_node(lin): 1})
            e.memo = lin
        return e.memo

    def reduce(self, e):
        """SOP node(s) for e = lin + sum x*y; returns a Lin of the stored results"""
        e = as_e(e)
        terms = list(e.terms)
        if not terms:
            return e.lin
        if e.lin.c:
            terms.append((self.as_lin(E(e.lin)), Lin({self.one: 1})))
        fixed = []
        for x, y in terms:
            if not x.encodable():
                x = Lin({self.lin_node(x): 1})
            if not y.encodable():
                y = Lin({self.lin_node(y): 1})
            if x.A() * y.A() > MAX_A_SUM:
                if x.A() >= y.A():
                    x = Lin({self.lin_node(x): 1})
                else:
                    y = Lin({self.lin_node(y): 1})
                if x.A() * y.A() > MAX_A_SUM:
                    x, y = Lin({self.lin_node(x): 1}), Lin({self.lin_node(y): 1})
            fixed.append((x, y))
        groups, cur, tot = [], [], 0
        for x, y in sorted(fixed, key=lambda t: -t[0].A() * t[1].A()):
            a = x.A() * y.A()
            if tot + a > MAX_A_SUM and cur:
                groups.append(cur)
                cur, tot = [], 0
            cur.append((x, y))
            tot += a
        groups.append(cur)
        out = Lin()
        for g in groups:
            vsum = sum(x.V() * y.V() for x, y in g)
            assert 1 + (vsum + V_DIV - 1) // V_DIV <= 1024, "value bound of the unreduced result"
            n = self._add(Node("sop", sum(x.val * y.val for x, y in g), 2))
            n.terms = g
            n.rv = vsum > V_DIV                        # T / R' < vsum p / 2520: at most p, so the Montgomery result is already below 2p
            n.level = max(max(x.level(), y.level()) for x, y in g) + 1
            out = out + Lin({n: 1})
        return out

    def fused(self, e, cv, s, cs):
        """ONE stored value cv * e + cs * s, s an already stored value: the linear step is the reducer's post-operation (64-bit
        per-limb arithmetic there, so |cv|, |cs| up to 12 need no headroom), no LIN round.  Falls back to store() when e does
        not fit one SOP node."""
        e, s = as_e(e), as_e(s)
        if s.terms or len(s.lin.c) != 1 or abs(list(s.lin.c.values())[0]) != 1:
            s = as_e(self.lin_node(self.reduce(s)))
        (sn, sk), = s.lin.c.items()
        cs *= sk                                           # a conjugated / negated stored value: the sign moves into the coefficient
        fresh = bool(e.terms)                              # only a node made by THIS reduction may take the post-operation
        lin = self.reduce(e)
        if not fresh or len(lin.c) != 1 or list(lin.c.values())[0] != 1 or list(lin.c)[0].kind != "sop" or list(lin.c)[0].post is not None:
            return self.store(E(lin.scale(cv) + Lin({sn: cs})))
        n = list(lin.c)[0]
        assert cv in COEF and cs in COEF and cv > 0
        n.post = (cv, sn, cs)
        n.val = (cv * n.val + cs * sn.val) % P
        n.level = max(n.level, sn.level + 1)
        return as_e(n)

    def store(self, e):
        """a stored value (single node) equal to e"""
        lin = self.reduce(e)
        return as_e(self.lin_node(lin))

    def inv(self, e):
        n = self.lin_node(self.reduce(e))
        m = self._add(Node("inv", pow(n.val, -1, P) if n.val else 0, 2))
        m.src = [(n, 1)]
        m.level = n.level + 1
        return as_e(m)

    def output(self, e, slot):
        n = self.lin_node(self.reduce(e))
        if n.kind in ("in", "const") or n.fixed_slot is not None:
            m = self._add(Node("sop", n.val, 2))                                  # a copy: value times one
            m.terms = [(Lin({n: 1}), Lin({self.one: 1}))]
            m.level = n.level + 1
            n = m
        n.fixed_slot = slot
        self.outputs.append(n)
        return n


G = None


# ---- Fp2 over E --------------------------------------------------------------------------------------------------------
class F2:
    __slots__ = ("re", "im")

    def __init__(self, re, im):
        self.re, self.im = as_e(re), as_e(im)

    @property
    def val(self):
        return (self.re.val, self.im.val)

    def __add__(self, o): return F2(self.re + o.re, self.im + o.im)
    def __sub__(self, o): return F2(self.re - o.re, self.im - o.im)
    def __neg__(self): return F2(-self.re, -self.im)
    def times(self, k): return F2(self.re.times(k), self.im.times(k))
    def conj(self): return F2(self.re, -self.im)
    def xi(self): return F2(self.re - self.im, self.re + self.im)               # times u + 1 (fp2.rs:156-166)

    def __mul__(self, o):                                                       # fp2.rs:205-222 as two sums of two products
        a0, a1, b0, b1 = G.as_lin(self.re), G.as_lin(self.im), G.as_lin(o.re), G.as_lin(o.im)
        na1, nb1 = a1.scale(-1), b1.scale(-1)                                   # - a1 b1: the sign goes where it costs less headroom
        neg = (na1, b1) if na1.A() * b1.A() <= a1.A() * nb1.A() else (a1, nb1)
        return F2(E(None, [(a0, b0), neg]), E(None, [(a0, b1), (a1, b0)]))

    def sqr(self):                                                              # fp2.rs:182-203
        a0, a1 = G.as_lin(self.re), G.as_lin(self.im)
        return F2(E(None, [(G.as_lin(E(a0 + a1)), G.as_lin(E(a0 + a1.scale(-1))))]), E(None, [(G.as_lin(E(a0.scale(2))), a1)]))

    def mul_fp(self, k):
        kk = G.as_lin(k)
        return F2(E(None, [(G.as_lin(self.re), kk)]), E(None, [(G.as_lin(self.im), kk)]))

    def stored(self):
        return F2(G.store(self.re), G.store(self.im))


def f2_fused(e, cv, s, cs):
    return F2(G.fused(e.re, cv, s.re, cs), G.fused(e.im, cv, s.im, cs))


def f2_const(c):
    return F2(G.cst(c[0]), G.cst(c[1]))


# ---- Fp12 in the basis w^k: e_k = c_{k mod 2}.c_{k div 2} -----------------------------------------------------------------
def f12_from_tower(f):           # oracle layout ((c0.c0, c0.c1, c0.c2), (c1.c0, c1.c1, c1.c2)) -> [e0..e5]
    return [f[k % 2][k // 2] for k in range(6)]


def f12_to_tower(e):
    return ((e[0], e[2], e[4]), (e[1], e[3], e[5]))


def f12_mul(a, b):
    out = []
    for k in range(6):
        acc = None
        for i in range(6):
            j = (k - i) % 6
            t = a[i] * b[j]
            if i + j >= 6:
                t = t.xi()
            acc = t if acc is None else acc + t
        out.append(acc)
    return out


def f12_sqr(a):
    out = []
    for k in range(6):
        acc = None
        for i in range(6):
            j = (k - i) % 6
            if i > j:
                continue
            t = a[i].sqr() if i == j else (a[i] * a[j]).times(2)
            if i + j >= 6:
                t = t.xi()
            acc = t if acc is None else acc + t
        out.append(acc)
    return out


def f12_mul_by_014(f, c0, c1, c4):
    """f * (c0 + c1 v + c4 v w): coefficients at w^0, w^2, w^3 (fp12.rs:116-128)"""
    sp = {0: c0, 2: c1, 3: c4}
    out = []
    for k in range(6):
        acc = None
        for j, c in sp.items():
            i = (k - j) % 6
            t = f[i] * c
            if i + j >= 6:
                t = t.xi()
            acc = t if acc is None else acc + t
        out.append(acc)
    return out


def f12_conj(f):
    return [f[k] if k % 2 == 0 else -f[k] for k in range(6)]


def f12_store(f):
    return [c.stored() for c in f]


def _fp2_pow(a, e):
    """a^e in Fp[u]/(u^2 + 1) on Python integers: the generator owns the arithmetic behind the constants it bakes into the program
    (the Frobenius coefficients of fp12.rs:149-168 / fp6.rs:159-185 are (u + 1)^(k (p^i - 1) / 6)); tests/ pin them to the oracle"""
    r = (1, 0)
    while e:
        if e & 1:
            r = ((r[0] * a[0] - r[1] * a[1]) % P, (r[0] * a[1] + r[1] * a[0]) % P)
        a = ((a[0] * a[0] - a[1] * a[1]) % P, 2 * a[0] * a[1] % P)
        e >>= 1
    return r


def f12_frobenius(f, power):
    """f^(p^power): conjugate the coefficients (odd powers) and scale e_k by (u + 1)^(k (p^power - 1) / 6)  (fp12.rs:145-171)"""
    out = []
    for k in range(6):
        c = f[k].conj() if power % 2 else f[k]
        g = _fp2_pow((1, 1), k * (P ** power - 1) // 6)
        out.append(c * f2_const(g) if k else c)
    return out


def f12_inv(f):
    """fp12.rs:187-194 with fp6.rs:294-312 and fp2.rs:300-319"""
    c0, c1 = [f[0], f[2], f[4]], [f[1], f[3], f[5]]

    def f6_mul(a, b):
        r = [None] * 3
        for i in range(3):
            for j in range(3):
                t = a[i] * b[j]
                if i + j >= 3:
                    t = t.xi()
                r[(i + j) % 3] = t if r[(i + j) % 3] is None else r[(i + j) % 3] + t
        return r
    s0, s1 = f6_mul(c0, c0), f6_mul(c1, c1)
    t = [s0[0] - s1[2].xi(), s0[1] - s1[0], s0[2] - s1[1]]              # c0^2 - v c1^2
    t = [x.stored() for x in t]
    A = t[0].sqr() - (t[1] * t[2]).xi()
    B = (t[2].sqr()).xi() - t[0] * t[1]
    C = t[1].sqr() - t[0] * t[2]
    A, B, C = A.stored(), B.stored(), C.stored()
    F = ((t[1] * C + t[2] * B).xi() + t[0] * A).stored()
    n = G.inv(F.re * F.re + F.im * F.im)
    Fi = F2(F.re * n, -(F.im * n)).stored()
    ti = [(A * Fi).stored(), (B * Fi).stored(), (C * Fi).stored()]
    r0, r1 = f6_mul(c0, ti), f6_mul(c1, [-x for x in ti])
    return [r0[0], r1[0], r0[1], r1[1], r0[2], r1[2]]


def cyclotomic_square(f, k=1):
    """k * f^2, pairings.rs:66-112 (Granger-Scott), on the cyclotomic subgroup.  fp4_square(a, b) = (xi b^2 + a^2, 2 a b); every
    new coefficient k (3 t -+ 2 z) is ONE sum of products with the linear step fused into its reduction: one round per squaring."""
    z0, z4, z3, z2, z1, z5 = f[0], f[2], f[4], f[1], f[3], f[5]

    def fp4(a, b):
        return b.sqr().xi() + a.sqr(), (a * b).times(2)
    t0, t1 = fp4(z0, z1)
    nz0 = f2_fused(t0, 3 * k, z0, -2 * k)
    nz1 = f2_fused(t1, 3 * k, z1, 2 * k)
    t0, t1 = fp4(z2, z3)
    t2, t3 = fp4(z4, z5)
    nz4 = f2_fused(t0, 3 * k, z4, -2 * k)
    nz5 = f2_fused(t1, 3 * k, z5, 2 * k)
    nz2 = f2_fused(t3.xi(), 3 * k, z2, 2 * k)
    nz3 = f2_fused(t2, 3 * k, z3, -2 * k)
    return [nz0, nz2, nz4, nz1, nz3, nz5]


def cyclotomic_square3(f, f3):
    """the same squaring inside a chain, where every value is kept TWICE: f and f3 = 3 f.  Then 3 t is a sum of products with
    one operand taken from the tripled copy (and 9 t with both), the "- 2 z" is one more product with the constant one, and
    every new coefficient -- of f^2 and of 3 f^2 -- is a PLAIN sum of products: no linear post-operation and no weak reduction
    on the reducer's path (the sum of |coefficients| stays within the column bound, which scaling by three would not)."""
    z = {0: f[0], 4: f[2], 3: f[4], 2: f[1], 1: f[3], 5: f[5]}
    w = {0: f3[0], 4: f3[2], 3: f3[4], 2: f3[1], 1: f3[3], 5: f3[5]}

    def fp4(a3, a, b3, b):                            # (3 or 9) * (xi b^2 + a^2, 2 a b), the factor carried by the first operands
        return (b3 * b).xi() + a3 * a, (a3 * b).times(2)

    def both(i, j):
        return fp4(w[i], z[i], w[j], z[j]), fp4(w[i], w[i], w[j], w[j])
    (t0, t1), (u0, u1) = both(0, 1)
    (t2, t3), (u2, u3) = both(2, 3)
    (t4, t5), (u4, u5) = both(4, 5)

    def outs(a, b, c, d, e, g, zz):
        nz0, nz1 = a - zz[0].times(2), b + zz[1].times(2)
        nz4, nz5 = c - zz[4].times(2), d + zz[5].times(2)
        nz2, nz3 = g.xi() + zz[2].times(2), e - zz[3].times(2)
        return [x.stored() for x in (nz0, nz2, nz4, nz1, nz3, nz5)]
    return outs(t0, t1, t2, t3, t4, t5, z), outs(u0, u1, u2, u3, u4, u5, w)


def cyclotomic_exp(f):
    """pairings.rs:114-132: f^|x| by square-and-multiply, conjugated"""
    if POST_CYCLOTOMIC:                               # experiment switch: the chain with the linear step as the reducer's post-operation
        tmp = f
        for b in reversed(range(63)):
            tmp = cyclotomic_square(tmp)
            if (BLS_X >> b) & 1:
                tmp = f12_store(f12_mul(tmp, f))
        return f12_conj(tmp)
    tmp, tmp3 = f, None
    for b in reversed(range(63)):                     # bit 63 is the leading one
        # a value that comes without its tripled copy (the chain's input, a product) is squared with the linear step as the
        # reducer's post-operation, once for f^2 and once for 3 f^2; inside a run of squarings both copies are plain sums
        tmp, tmp3 = cyclotomic_square3(tmp, tmp3) if tmp3 is not None else (cyclotomic_square(tmp), cyclotomic_square(tmp, 3))
        if (BLS_X >> b) & 1:
            tmp, tmp3 = f12_store(f12_mul(tmp, f)), None
    return f12_conj(tmp)


def final_exponentiation(f):
    """pairings.rs:134-173"""
    t0 = f12_conj(f)                                  # six Frobenius maps
    t1 = f12_store(f12_inv(f))
    t2 = f12_store(f12_mul(t0, t1))
    t1 = t2
    t2 = f12_store(f12_mul(f12_frobenius(t2, 2), t1))
    t1 = f12_conj(cyclotomic_square(t2))
    t3 = cyclotomic_exp(t2)
    t4 = cyclotomic_square(t3)
    t5 = f12_store(f12_mul(t1, t3))
    t1 = cyclotomic_exp(t5)
    t0 = cyclotomic_exp(t1)
    t6 = cyclotomic_exp(t0)
    t6 = f12_store(f12_mul(t6, t4))
    t4 = cyclotomic_exp(t6)
    t5 = f12_conj(t5)
    t4 = f12_store(f12_mul(t4, f12_store(f12_mul(t5, t2))))
    t5 = f12_conj(t2)
    t1 = f12_store(f12_mul(t1, t2))
    t1 = f12_store(f12_frobenius(t1, 3))
    t6 = f12_store(f12_mul(t6, t5))
    t6 = f12_store(f12_frobenius(t6, 1))
    t3 = f12_store(f12_mul(t3, t0))
    t3 = f12_store(f12_frobenius(t3, 2))
    t3 = f12_store(f12_mul(t3, t1))
    t3 = f12_store(f12_mul(t3, t6))
    return f12_mul(t3, t4)


def miller_loop(px, py, qx, qy):
    """pairings.rs:668-770 (the unprepared loop of `pairing`, :607-653), line coefficients as values: the doubling step's
    (tmp0, tmp3, tmp6) are 4 y z^3, -6 x^2 z^2 and 6 x^3 - 4 y^2 of the CURRENT point, the new point (9x^4 - 8xy^2, ..., 2yz)."""
    x, y, z = qx, qy, F2(G.cst(1), G.cst(0))
    f = None

    def ell(f, la, lb, lc):
        c4 = la.mul_fp(py).stored()
        c1 = lb.mul_fp(px).stored()
        lc = lc.stored()
        if f is None:                                # f = 1: the product is the line itself
            zero = F2(G.cst(0), G.cst(0))
            return [lc, zero, c1, c4, zero, zero]
        return f12_store(f12_mul_by_014(f, lc, c1, c4))

    def doubling():
        """three levels per step (the critical chain is y -> y^2 -> new x, new y), every value a PLAIN sum of products -- scaled
        copies (3x^2, 3u, 4y^2, 4y^4) are separate sums or lazy multiples, so no reducer carries a linear post-operation:
        level 1  3 x^2 (schoolbook, the factor on one operand), y^2, y z, z^2, 8 x;
        level 2  x' = (3x^2)^2 - 8x y^2,  u = 3 x^3 and 3u,  4 y^4,  4 y^2,  the line's  4 y z^3  and  -6 x^2 z^2;
        level 3  y' = 3u (4y^2 - u) - 8 y^4  written as  3u * 4y^2 - 3u * u - 2 * 4y^4,   6 x^3 - 4 y^2.
        z' = 2 y z stays a lazy multiple.  Values are the reference's (pairings.rs:709-738)."""
        nonlocal x, y, z
        a0, a1 = G.as_lin(x.re), G.as_lin(x.im)
        xx3 = F2(E(None, [(a0.scale(3), a0), (a1.scale(-3), a1)]), E(None, [(a0.scale(6), a1)])).stored()
        yy, zz, yz = y.sqr().stored(), z.sqr().stored(), (y * z).stored()
        x8 = x.times(8).stored()
        nx = (xx3.sqr() - x8 * yy).stored()           # 9 x^4 - 8 x y^2
        u, u3 = (x * xx3).stored(), (x.times(3) * xx3).stored()
        b0, b1 = G.as_lin(yy.re), G.as_lin(yy.im)
        y4x4 = F2(E(None, [(b0.scale(4), b0), (b1.scale(-4), b1)]), E(None, [(b0.scale(8), b1)])).stored()
        yy4 = yy.times(4).stored()
        xxzz3, yzzz = (xx3 * zz).stored(), (yz * zz).stored()
        lc = (u.times(2) - yy4).stored()              # 6 x^3 - 4 y^2
        ny = (u3 * yy4 - u3 * u - y4x4.times(2)).stored()
        la, lb = yzzz.times(4), -xxzz3.times(2)
        x, y, z = nx, ny, yz.times(2)
        return la, lb, lc

    def addition():
        nonlocal x, y, z
        zsq, ysq = z.sqr().stored(), qy.sqr().stored()
        t0 = (zsq * qx).stored()
        t1 = ((((qy + z).sqr() - ysq - zsq)) * zsq).stored()
        t2 = (t0 - x).stored()
        t3 = t2.sqr().stored()
        t4 = t3.times(4)
        t5 = (t4 * t2).stored()
        t6 = (t1 - y - y).stored()
        t9 = (t6 * qx).stored()
        t7 = (t4 * x).stored()
        nx = (t6.sqr() - t5 - t7 - t7).stored()
        nz = ((z + t2).sqr() - zsq - t3).stored()
        t10 = qy + nz
        t8 = ((t7 - nx) * t6)
        t0b = (y * t5)
        ny = (t8 - t0b.times(2)).stored()
        t10b = t10.sqr() - ysq - nz.sqr()
        lc = t9.times(2) - t10b
        la, lb = nz.times(2), -t6.times(2)
        x, y, z = nx, ny, nz
        return la, lb, lc

    bits = []
    found = False
    for b in reversed(range(64)):
        i = ((BLS_X >> 1) >> b) & 1
        if not found:
            found = bool(i)
            continue
        bits.append(i)
    for bit in bits:
        f = ell(f, *doubling())
        if bit:
            f = ell(f, *addition())
        f = f12_store(f12_sqr(f))
    f = ell(f, *doubling())
    return f12_conj(f)                                # BLS_X_IS_NEGATIVE


# ---- scheduling, slot allocation, encoding -------------------------------------------------------------------------------
def sources(n):
    if n.kind == "sop":
        return [m for x, y in n.terms for m in list(x.c) + list(y.c)] + ([n.post[1]] if n.post else [])
    return [m for m, _ in n.src]


def schedule(g, nfixed):
    """rounds (ASAP levels, split when a level needs more lanes than there are) and LDS slots by liveness.
    Slots [0, nfixed) belong to inputs / outputs and are never handed out as scratch; constants follow."""
    need, stack = set(), list(g.outputs)
    while stack:
        n = stack.pop()
        if n not in need:
            need.add(n)
            if n.kind in ("sop", "lin", "inv"):
                stack.extend(sources(n))
    byl = {}
    for n in g.nodes:
        if n in need and n.kind in ("sop", "lin", "inv"):
            byl.setdefault(n.level, []).append(n)
    rounds = []
    for lv in sorted(byl):
        cur, lanes = [], 0
        nchunk = (NL + CHUNK - 1) // CHUNK
        nacc = 0
        for n in byl[lv]:
            w = len(n.terms) * nchunk if n.kind == "sop" else 1
            if lanes + w > LANES - 64 or (n.kind == "sop" and nacc == MAX_ACC):     # 64: room to start the LIN / INV lanes on a wavefront boundary
                rounds.append(cur)
                cur, lanes, nacc = [], 0, 0
            cur.append(n)
            lanes += w
            nacc += n.kind == "sop"
        rounds.append(cur)
    rnd_of = {n: r for r, ns in enumerate(rounds) for n in ns}
    last = {}
    for n, r in rnd_of.items():
        for m in sources(n):
            last[m] = max(last.get(m, -1), r)
        if n.kind == "sop" and n.post:                      # the post-operation reads its slot in the WRITE phase of the round:
            last[n.post[1]] = max(last.get(n.post[1], -1), r + 1)        # nobody may write that slot before the next round
    consts = [c for c in g.consts.values() if c in need]
    for i, c in enumerate(consts):
        c.slot = nfixed + i
    for n in g.inputs:
        n.slot = n.fixed_slot
    base = nfixed + len(consts)
    busy = {}                                          # scratch slot -> round of its occupant's last read
    nslots = base
    for r, ns in enumerate(rounds):
        free = sorted(s for s, until in busy.items() if until <= r)       # reads of a round precede its writes
        for n in ns:
            if n.fixed_slot is not None:
                n.slot = n.fixed_slot
                for i in g.inputs:                       # an output may take an input's slot only after that input's last read
                    assert i.slot != n.slot or last.get(i, -1) <= r, "output slot still holds a live input"
                continue
            if free:
                n.slot = free.pop(0)
            else:
                n.slot = nslots
                nslots += 1
            busy[n.slot] = last.get(n, r)
    return rounds, consts, nslots


def bias_limbs():
    """fe.hip.h make_bias(VB, 1): VB * p spread over the limbs so that every limb is at least 2^28 - 1"""
    pl = [(P >> (LW * i)) & ((1 << LW) - 1) for i in range(NL)]
    r, carry = [0] * NL, 0
    for i in range(NL):
        t = pl[i] * VB + carry
        r[i] = (t & ((1 << LW) - 1)) if i < NL - 1 else t
        carry = (t >> LW) if i < NL - 1 else 0
    r[0] += 1 << LW
    for i in range(1, NL - 1):
        r[i] += (1 << LW) - 1
    r[NL - 1] -= 1
    assert sum(v << (LW * i) for i, v in enumerate(r)) == VB * P
    return r


def enc_pairs(items):
    """operand word: up to two (slot 8 bits, signed 5-bit coefficient) pairs at bits 0 and 13, the weight of the bias slot
    (the sum of the negative coefficients' magnitudes) at bit 26"""
    assert 1 <= len(items) <= 2, len(items)
    w, wneg = 0, 0
    for i, (n, k) in enumerate(items):
        assert n.slot is not None and n.slot < MAX_SLOTS and -16 <= k <= 15 and k, (n.kind, n.slot, k)
        w |= (n.slot | ((k & 31) << 8)) << (13 * i)
        wneg += -k if k < 0 else 0
    assert wneg <= 15
    return w | (wneg << 26)


def enc_lin(lin):
    return enc_pairs(sorted(lin.c.items(), key=lambda t: t[0].slot))


def encode(g, rounds, consts, nslots, n_in, n_out):
    """binary layout (u32 words):
       header[16]: magic, nrounds, nslots, nconst, n_in, n_out, const_off, rounds_off, pdesc_off, rdesc_off, lanes, chunk, bias_slot, max_acc
       consts:     nconst x (slot, 14 limbs)  -- c * R' mod p, and the spread VB * p the negative coefficients lean on
       rounds:     nrounds x (first product descriptor, product lanes, first reducer descriptor, reducer lanes)
       pdesc:      4 words per lane: w0 = op | lo << 2 | len << 6 | accumulator << 10 ; w1, w2 = operand words ; w3 = out slot (LIN, INV)
       rdesc:      2 words per reducer lane: w0 = 1 | accumulator << 1 | out << 11 | post << 19 | post_slot << 20 | weak_reduce << 28 ;
                   w1 = cv (5 bits, signed) | cs << 5"""
    nchunk = (NL + CHUNK - 1) // CHUNK
    pdescs, rdescs, rtab = [], [], []
    bias_slot = nslots
    nslots += 1
    assert nslots <= MAX_SLOTS
    max_acc = 0
    for ns in rounds:
        pfirst, rfirst = len(pdescs), len(rdescs)
        sops = [n for n in ns if n.kind == "sop"]
        others = [n for n in ns if n.kind != "sop"]
        assert len(sops) <= MAX_ACC
        max_acc = max(max_acc, len(sops))
        lanes = []
        # term-major: the lanes that add into the same columns of one accumulator sit len(sops) * nchunk lanes apart
        for t in range(max([len(n.terms) for n in sops] + [0])):
            for a, n in enumerate(sops):
                if t >= len(n.terms):
                    continue
                x, y = n.terms[t]
                for c in range(nchunk):
                    lo = c * CHUNK
                    ln = min(CHUNK, NL - lo)
                    lanes.append([OP_SOP | (lo << 2) | (ln << 6) | (a << 10), enc_lin(x), enc_lin(y), 0])
        for a, n in enumerate(sops):
            w0 = 1 | (a << 1) | (n.slot << 11) | ((1 if n.rv else 0) << 28)
            w1 = 0
            if n.post:
                cv, sn, cs = n.post
                assert 0 < cv <= 15 and -16 <= cs <= 15
                w0 |= (1 << 19) | (sn.slot << 20)
                w1 = (cv & 31) | ((cs & 31) << 5)
            rdescs.append([w0, w1])
        if others and len(lanes) % 64 and len(lanes) + 64 - len(lanes) % 64 + len(others) <= LANES:
            lanes += [[0, 0, 0, 0]] * (64 - len(lanes) % 64)          # LIN / INV lanes start a wavefront of their own (different code path)
        for n in others:
            if n.kind == "lin":
                items = n.src
                assert 1 <= len(items) <= 4
                lanes.append([OP_LIN, enc_pairs(items[0:2]), enc_pairs(items[2:4]) if len(items) > 2 else 0, n.slot])
            else:
                lanes.append([OP_INV, enc_pairs(n.src), 0, n.slot])
        assert len(lanes) <= LANES, len(lanes)
        pdescs += lanes
        rtab.append((pfirst, len(lanes), rfirst, len(sops)))
    pdescs.append([0, 0, 0, 0])                                       # the prefetch of the round after the last reads index 0 of nothing: keep it inside the tables
    rdescs.append([0, 0])
    hdr = [0x57494432, len(rounds), nslots, len(consts) + 1, n_in, n_out, 0, 0, 0, 0, LANES, CHUNK, bias_slot, max_acc, 0, 0]
    const_words = []
    for c in consts:
        v = c.val * RP % P
        const_words += [c.slot] + [(v >> (LW * i)) & ((1 << LW) - 1) for i in range(NL)]
    const_words += [bias_slot] + bias_limbs()
    const_words += [0] * (-(16 + len(const_words)) % 4)              # the round table is read as 16-byte quads
    round_words = [w for r in rtab for w in r]
    pwords = [w for d in pdescs for w in d]
    rwords = [w for d in rdescs for w in d]
    hdr[6] = 16
    hdr[7] = 16 + len(const_words)
    hdr[8] = hdr[7] + len(round_words)
    hdr[9] = hdr[8] + len(pwords)
    allw = hdr + const_words + round_words + pwords + rwords
    allw += [0] * (-len(allw) % 4)
    return struct.pack("<%dI" % len(allw), *allw)


def build_programs(check=False):
    """(miller, final_exp) encoded programs.  The symbolic nodes carry their value for ONE sample input (SAMPLE_P, SAMPLE_Q below: the
    generator's own literals) so that constant folding and the bound bookkeeping can be asserted while the program is built; with
    check=True (tests/ only: the product build never passes it) the values are also compared with oracle/bls12_381_ref.py"""
    global G
    Pa, Qa = SAMPLE_P, SAMPLE_Q
    assert (Pa[1] * Pa[1] - Pa[0] ** 3 - 4) % P == 0, "SAMPLE_P is not on y^2 = x^3 + 4"
    if check:
        sys.path.insert(0, ROOT)
        from oracle import bls12_381_ref as o         # test infrastructure: reached only through --check
        r = o.SplitMix64(2026)
        assert o.g1_to_affine(o.g1_affine_mul(o.G1_GEN, r.scalar()))[:2] == SAMPLE_P, "SAMPLE_P is not [k]G1 of the oracle's stream"
        assert o.g2_to_affine(o.g2_affine_mul(o.G2_GEN, r.scalar()))[:2] == SAMPLE_Q, "SAMPLE_Q is not [k]G2 of the oracle's stream"
        for k in range(6):
            for power in (1, 2, 3):
                assert _fp2_pow((1, 1), k * (P ** power - 1) // 6) == tuple(c % P for c in o.fp2_pow((1, 1), k * (P ** power - 1) // 6))
    progs = {}
    # Miller loop: inputs px py qx.re qx.im qy.re qy.im in slots 12..17, outputs (tower order c0.c0.re ... c1.c2.im) in slots 0..11
    G = Graph()
    px, py = G.inp(Pa[0], 12), G.inp(Pa[1], 13)
    qx, qy = F2(G.inp(Qa[0][0], 14), G.inp(Qa[0][1], 15)), F2(G.inp(Qa[1][0], 16), G.inp(Qa[1][1], 17))
    f = miller_loop(px, py, qx, qy)
    tower = f12_to_tower(f)
    flat = [c for half in tower for fp2 in half for c in (fp2.re, fp2.im)]
    for i, c in enumerate(flat):
        G.output(c, i)
    if check:
        want = o.fp12_flatten(o.miller_loop(Pa + (False,), Qa + (False,)))
        assert [n.val for n in G.outputs] == [w % P for w in want], "Miller program differs from the oracle"
    ml_val = [n.val for n in G.outputs]
    rounds, consts, nslots = schedule(G, 18)
    progs["miller"] = (encode(G, rounds, consts, nslots, 18, 12), len(rounds), nslots, len(G.nodes))
    # final exponentiation: inputs = outputs = slots 0..11
    G = Graph()
    ins = [G.inp(v, i) for i, v in enumerate(ml_val)]
    tw = ((F2(ins[0], ins[1]), F2(ins[2], ins[3]), F2(ins[4], ins[5])), (F2(ins[6], ins[7]), F2(ins[8], ins[9]), F2(ins[10], ins[11])))
    f = final_exponentiation(f12_from_tower(tw))
    tower = f12_to_tower(f)
    flat = [c for half in tower for fp2 in half for c in (fp2.re, fp2.im)]
    for i, c in enumerate(flat):
        G.output(c, i)
    if check:
        want = o.fp12_flatten(o.pairing(Pa + (False,), Qa + (False,)))
        assert [n.val for n in G.outputs] == [w % P for w in want], "final exponentiation program differs from the oracle"
    rounds, consts, nslots = schedule(G, 12)
    progs["final_exp"] = (encode(G, rounds, consts, nslots, 12, 12), len(rounds), nslots, len(G.nodes))
    return progs


# the sample input the node values are carried for: affine coordinates of one point of G1 and one of G2 (subgroup points, so that
# the checked run can compare with the pairing of the oracle); literals, so that the generator needs nothing outside this file
SAMPLE_P = (2953704839879994580174565036682715916180523479062088721756859108067000074983573215055813609902719574827537402469893,
            2899222940235106533380767082349259515979568556411165493167315793440371091908654876618364418118559910748204779541058)
SAMPLE_Q = ((731565187646367782085872659530889344490627462904754225030790530923945044183985844094010590147024639760215747998007,
             3945096305268708607333280870131116412615992429647936549067354213625000321693642968900745670590454298583381806024262),
            (2142671969214379813623204901809485091240339275669814619282785327695981863376685360930435794692371520888022959880221,
             786019518877564554804028913490236040290455969355151147466247272841717321863948206830577347573464900792183413848521))
FORMAT_VERSION = 4                        # word 15 of the blob header; wide.hip.h WIDE_FORMAT_VERSION must agree (bump with any change of the encoding)

CONFIGS = [(1024, 4), (512, 8)]           # (lanes, limbs of x per product lane) built into the library: lowest latency / two workgroups per CU


def main():
    """blob: 16-word header [magic, number of programs, (offset, length) x programs] then, for every configuration in CONFIGS
    (or the single --lanes / --chunk one), the Miller-loop and the final-exponentiation program"""
    global LANES, CHUNK, POST_CYCLOTOMIC
    POST_CYCLOTOMIC = "--post-cyclotomic" in sys.argv
    out = os.path.join(ROOT, "bls12_381_amd", "wide_prog.bin")
    if "--out" in sys.argv:
        out = sys.argv[sys.argv.index("--out") + 1]
    configs = CONFIGS
    if "--lanes" in sys.argv or "--chunk" in sys.argv:
        configs = [(int(sys.argv[sys.argv.index("--lanes") + 1]) if "--lanes" in sys.argv else LANES,
                    int(sys.argv[sys.argv.index("--chunk") + 1]) if "--chunk" in sys.argv else CHUNK)]
    blob = b""
    index = []
    off = 16 * 4
    for LANES, CHUNK in configs:
        progs = build_programs(check="--check" in sys.argv)
        for name in ("miller", "final_exp"):
            data, nr, ns, nn = progs[name]
            index += [off // 4, len(data) // 4]
            off += len(data)
            blob += data
            print("%4d x %d  %-10s %5d rounds, %4d slots, %6d nodes, %7d bytes" % (LANES, CHUNK, name, nr, ns, nn, len(data)), file=sys.stderr)
    assert len(index) <= 13
    head = struct.pack("<16I", 0x57504752, len(index) // 2, *index, *([0] * (13 - len(index))), FORMAT_VERSION)
    tmp = "%s.tmp.%d" % (out, os.getpid())        # several ranks may build at once: a reader never sees a half-written file
    with open(tmp, "wb") as fh:
        fh.write(head + blob)
    os.replace(tmp, out)
    assert "--check" in sys.argv or not any(m == "oracle" or m.startswith("oracle.") for m in sys.modules), "the generator reached the oracle without --check"
    print("wrote", out, len(head) + len(blob), "bytes", file=sys.stderr)


if __name__ == "__main__":
    main()
