import json,sys
for f in sys.argv[1:]:
    d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    r=d['roofline']
    print(f.split('/')[-1], 'value %.4g step %.3f | launch_ms %.3f frac %.3f iso %.3f | whole pipelined %.3f' % (d['value'], d['ms_per_step'], r['launch_ms'], r['frac'], r['frac_isolated'], r['whole_msm_frac_pipelined']))
