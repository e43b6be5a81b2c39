import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{"metric"'):
        d=json.loads(l)
        r=d['roofline']
        print(sys.argv[1], 'headline', d['value'], d['ms_per_step'], r.get('mean_sclk_mhz'), r.get('mean_power_w'))
        for k,v in (d.get('extra') or {}).items():
            if 'value' in v: print('   ', k, v['value'], v['ms_per_step'], v['roofline'].get('mean_sclk_mhz'), v['roofline'].get('mean_power_w'))
