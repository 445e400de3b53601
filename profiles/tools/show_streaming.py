import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['streaming']
for name,x in (('packed16 pinned', d), ('stride32 pinned', d['stride32']), ('stride32 pageable -> packed', d['stride32_packed'])):
    print(name, round(x['frames_per_s']), 'frames/s', round(x['h2d_GBps'],1), 'GB/s', x.get('pack_threads'), x.get('packed_equals_source'))
