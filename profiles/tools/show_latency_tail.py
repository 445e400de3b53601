import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['latency']
def show(name,x):
    print(name, 'median', round(x['ms_per_frame_median']*1e3,1), 'p99', round(x['ms_per_frame_p99']*1e3,1))
    print('   median phases', {k:round(v,1) for k,v in x['host_us_median'].items()})
    print('   p99-call phases', {k:(round(v,1) if isinstance(v,float) else v) for k,v in x['host_us_p99_calls'].items()})
show('supplied', d); show('ransac', d['estimated']['ransac']); show('semantic', d['estimated']['semantic'])
