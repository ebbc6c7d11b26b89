#!/bin/bash
# static instruction mix of the S=8 kernels (proxy for the VALU-bound batched throughput)
cd /root/repo/hc-mvs_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -DHCMVS_COUNT -S --cuda-device-only -o /tmp/pm_kernels.s pm_kernels.hip 2>/dev/null
python3 - <<'PY'
import collections
lines=open('/tmp/pm_kernels.s').read().split('\n')
starts=[(i,l.split(':')[0]) for i,l in enumerate(lines) if l.startswith('_ZN5hcmvs') and ':' in l]
for n,(i,name) in enumerate(starts):
    if 'sweep_kernelILi8ELi1' in name or 'probe_' in name or 'score_kernelILi8' in name:
        end=starts[n+1][0] if n+1<len(starts) else len(lines)
        ins=[l.strip().split()[0] for l in lines[i:end] if l.startswith('\t') and l.strip() and not l.strip().startswith(('.',';'))]
        c=collections.Counter('valu' if x.startswith('v_') else 'salu' if x.startswith('s_') else 'mem' for x in ins)
        vg=[l for l in lines[i:end+400] if '.vgpr_count' in l or 'NumVgprs' in l]
        print(name[9:44], len(ins), dict(c))
PY
