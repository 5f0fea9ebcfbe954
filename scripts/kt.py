import json,sys
d=json.load(open(sys.argv[1]))
print(sys.argv[1], "ms/step %.3f" % d['ms_per_step'], {k: round(v['avg_ms'],3) for k,v in d['kernels'].items() if v['avg_ms']>0.1})
