#!/bin/bash
# usage: scratch/r5/res.sh <file.hip> [grep-pattern] [extra hipcc flags...]   -- per-kernel registers / spills / LDS of one unit
f=$1; pat=${2:-.}; shift; shift
cd /root/repo/chadavit_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops "$@" \
  -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/res_$$.o 2>&1 | python3 -c "
import sys,re,subprocess
cur={}
rows=[]
for ln in sys.stdin:
    m=re.search(r'remark: +(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)',ln)
    if not m:
        if 'error' in ln or 'warning' in ln: print(ln.rstrip())
        continue
    k,v=m.groups()
    if k=='Function Name':
        cur={'name':v}; rows.append(cur)
    else: cur[k]=v
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    n=re.sub(r'\(anonymous namespace\)::','',n); n=re.sub(r'\(.*','',n)
    print(f\"{n:60s} v={r.get('VGPRs')} a={r.get('AGPRs')} s={r.get('TotalSGPRs')} scratch={r.get('ScratchSize [bytes/lane]')} vspill={r.get('VGPRs Spill')} occ={r.get('Occupancy [waves/SIMD]')} lds={r.get('LDS Size [bytes/block]')}\")
" | grep -E "$pat"
rm -f /tmp/res_$$.o
