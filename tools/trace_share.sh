#!/bin/bash
# Kernel timeline of one rank's 1/8 share in the default (pipelined) mode:  tools/trace_share.sh <workload> [share]
set -u
W=${1:-c2}; SH=${2:-0/8}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/trace_share_$W
rm -rf "$OUT"; mkdir -p "$OUT"
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -- python3 bench.py --workload $W --steps 40 --warmup 5 --no-cpu-baseline --rehearse-share $SH > "$OUT/log" 2>&1 || { echo trace failed; tail -5 "$OUT/log"; exit 1; }
find "$OUT/t" -name '*kernel_trace.csv' -exec cp {} "$OUT/trace.csv" \;
python3 - "$OUT/trace.csv" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last 25 frames: take the last 100 render kernels
rk=[r for r in rows if 'render_kernel' in r['Kernel_Name'] and 'Lb1ELb1' not in r['Kernel_Name']]
rk=rk[-60:]
t0=int(rk[0]['Start_Timestamp']); t1=int(rk[-1]['End_Timestamp'])
span=(t1-t0)/1e3
print(f"{len(rk)} render kernels over {span:.0f} us -> {span/len(rk):.1f} us per frame")
dur=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rk]
print("render_kernel duration us: min %.0f avg %.0f max %.0f"%(min(dur),sum(dur)/len(dur),max(dur)))
names=collections.defaultdict(list)
for r in rows:
    if int(r['Start_Timestamp'])>=t0 and int(r['End_Timestamp'])<=t1:
        n=r['Kernel_Name'].split('(')[0].split('<')[0].split('::')[-1]
        names[n].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for n,v in names.items(): print(f"  {n:24s} n={len(v):4d} avg {sum(v)/len(v):8.1f} us")
# concurrency: how many render kernels overlap on average
ev=[]
for r in rk: ev+= [(int(r['Start_Timestamp']),1),(int(r['End_Timestamp']),-1)]
ev.sort(); c=0; last=ev[0][0]; acc=collections.Counter()
for t,d in ev:
    acc[c]+=t-last; last=t; c+=d
tot=sum(acc.values())
print("render kernels in flight (share of time):", {k:round(v/tot,2) for k,v in sorted(acc.items())})
# gaps per queue between resolve end and next probe start
q=collections.defaultdict(list)
for r in rows:
    if int(r['Start_Timestamp'])>=t0: q[r['Queue_Id']].append(r)
for k,v in q.items():
    seq=[(x['Kernel_Name'].split('(')[0].split('<')[0].split('::')[-1],int(x['Start_Timestamp']),int(x['End_Timestamp'])) for x in v]
    gaps=[(seq[i+1][1]-seq[i][2])/1e3 for i in range(len(seq)-1)]
    if gaps: print(f"  queue {k}: {len(seq)} kernels, gap between consecutive kernels avg {sum(gaps)/len(gaps):.1f} us max {max(gaps):.1f}")
PY
