import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f,'FAILED',e); continue
    e=d['extras']
    print(f, 'value %.4g step %.3f single %.3f | g2 %.4g (%.2f ms, single %.2f) | pair %.4g mml %.4g | ntt %.4f ms | h2c_g2 %.4g | tables %.3f ms' % (d['value'], d['ms_per_step'], d['single_call_ms'], e['g2_msm_scalar_muls_per_s'], e['g2_msm']['ms'], e['g2_msm']['single_call_ms'], e['pairings_per_s'], e['multi_miller_loop_terms_per_s'], e['fr_ntt']['ms'], e['hash_to_g2']['hashes_per_s'], e['g1_msm_precomputed_tables']['ms']))
