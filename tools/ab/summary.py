import json,glob,sys,statistics
d=sys.argv[1]
for v in sys.argv[2:]:
    ms=[];ro=[];ph={}
    for f in sorted(glob.glob(f'{d}/{v}.*.json')):
        try: b=json.load(open(f))
        except Exception as e: print(f, 'unreadable'); continue
        ms.append(b['ms_per_step']); ro.append(b['reference_order']['ms_per_step'])
        for k,x in b['phases_ms'].items(): ph.setdefault(k,[]).append(x)
    print(v, 'headline', [round(x,4) for x in ms], 'median', round(statistics.median(ms),4), '| ref-order', [round(x,4) for x in ro])
    print('   phases', {k: round(statistics.median(x),4) for k,x in ph.items()})
