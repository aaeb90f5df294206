#!/bin/bash
# Runs on the GPU box: kernel trace of the two-context probe; how much of the shade kernels' time runs while an extend kernel is running.
export TMPDIR=/tmp
out=gpurun_out/overlap
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 tools/two_contexts_probe.py > $out/probe.txt 2> $out/err.txt
python3 - $out <<'PY'
import csv,glob,sys
rows=[]
for f in glob.glob(sys.argv[1]+'/trace/*/*_kernel_trace.csv'):
    for r in csv.DictReader(open(f)): rows.append((r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Queue_Id'), r.get('Stream_Id')))
ext=[(s,e,q) for n,s,e,q,_ in rows if 'k_extend<' in n]
sh=[(s,e,q) for n,s,e,q,_ in rows if 'k_finish_shade' in n]
print(len(ext),'extend',len(sh),'shade kernels; queues', sorted(set(q for _,_,q in ext)))
tot=0; ov=0
for s,e,q in sh:
    tot+=e-s
    for s2,e2,q2 in ext:
        lo=max(s,s2); hi=min(e,e2)
        if hi>lo and q2!=q: ov+=hi-lo
print('shade time %.1f ms, of which %.1f ms while another queue runs extend' % (tot/1e6, ov/1e6))
import statistics
print('shade avg ms', statistics.mean((e-s)/1e6 for s,e,_ in sh), 'extend avg ms', statistics.mean((e-s)/1e6 for s,e,_ in ext))
PY
tail -3 $out/probe.txt
