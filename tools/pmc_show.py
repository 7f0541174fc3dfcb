import csv, collections, sys
rows = list(csv.DictReader(open('gpurun_out/pmc_counters.csv')))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r['Kernel_Name'].replace('void ', '')
    if not n.startswith('k_'): continue
    k = n.split('(')[0]
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    agg[k]['dur_us'].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    agg[k]['lds'].append(int(r['LDS_Block_Size'])); agg[k]['vgpr'].append(int(r['VGPR_Count']))
for k,v in sorted(agg.items(), key=lambda kv: -sum(kv[1]['dur_us'])/len(kv[1]['dur_us'])):
    m = {c: sum(x)/len(x) for c,x in v.items()}
    w = max(m.get('SQ_WAVES',1),1)
    print(f"{k:20s} dur={m['dur_us']:8.1f}us lds={int(m['lds']):6d} vgpr={int(m['vgpr']):4d} waves={int(w):8d} valu/w={m.get('SQ_INSTS_VALU',0)/w:8.0f} salu/w={m.get('SQ_INSTS_SALU',0)/w:8.0f} lds/w={m.get('SQ_INSTS_LDS',0)/w:6.0f} vmem/w={m.get('SQ_INSTS_VMEM_RD',0)/w:6.0f} cyc/w={m.get('SQ_WAVE_CYCLES',0)/w*4:9.0f} wait%={100*m.get('SQ_WAIT_ANY',0)/max(m.get('SQ_WAVE_CYCLES',1),1):5.1f}")
