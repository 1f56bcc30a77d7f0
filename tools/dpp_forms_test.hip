#include <hip/hip_runtime.h>
#include <stdio.h>
#define DPP " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
__global__ void k(const unsigned* in, unsigned* out) {
  unsigned x = in[threadIdx.x], y = in[64 + threadIdx.x];
  unsigned r[8];
  asm volatile("s_nop 7\n v_add_u32_dpp %0, %1, %2" DPP : "=&v"(r[0]) : "v"(x), "v"(y));
  asm volatile("s_nop 7\n v_sub_u32_dpp %0, %1, %2" DPP : "=&v"(r[1]) : "v"(x), "v"(y));
  asm volatile("s_nop 7\n v_subrev_u32_dpp %0, %1, %2" DPP : "=&v"(r[2]) : "v"(x), "v"(y));
  asm volatile("s_nop 7\n v_and_b32_dpp %0, %1, %2" DPP : "=&v"(r[3]) : "v"(x), "v"(y));
  asm volatile("s_nop 7\n v_mov_b32_dpp %0, %1" DPP : "=&v"(r[4]) : "v"(x));
  asm volatile("s_nop 7\n v_xor_b32_dpp %0, %1, %2" DPP : "=&v"(r[5]) : "v"(x), "v"(y));
  asm volatile("s_nop 7\n v_lshlrev_b32_dpp %0, %1, %2" DPP : "=&v"(r[6]) : "v"(x & 7), "v"(y));
  unsigned z = x;
  asm volatile("s_nop 7\n v_add_u32_dpp %0, %1, %0" DPP : "+v"(z) : "v"(y));
  r[7] = z;
  for (int i = 0; i < 8; i++) out[threadIdx.x * 8 + i] = r[i];
}
int main() {
  unsigned h[128], *di, *dout, ho[512];
  for (int i = 0; i < 128; i++) h[i] = (i * 2654435761u + 12345) & 0xfffffff;
  (void)hipMalloc(&di, 512); (void)hipMalloc(&dout, 2048);
  (void)hipMemcpy(di, h, 512, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  (void)hipMemcpy(ho, dout, 2048, hipMemcpyDeviceToHost);
  const char* nm[8] = {"add", "sub", "subrev", "and", "mov", "xor", "lshlrev", "add_inplace"};
  int bad[8] = {0}, alt[8] = {0};
  for (int i = 0; i < 64; i++) {
    unsigned x = h[i], px = h[i ^ 1], y = h[64 + i], py = h[64 + (i ^ 1)];
    unsigned want[8] = {px + y, px - y, y - px, px & y, px, px ^ y, y << (px & 7), py + x};
    unsigned other[8] = {x + py, x - py, py - x, x & py, px, x ^ py, py << (x & 7), y + px};
    for (int v = 0; v < 8; v++) { if (ho[i * 8 + v] != want[v]) bad[v]++; if (ho[i * 8 + v] == other[v]) alt[v]++; }
  }
  for (int v = 0; v < 8; v++) printf("%-12s wrong lanes %2d   (matches 'dpp on the other operand' in %2d lanes)\n", nm[v], bad[v], alt[v]);
  return 0;
}
