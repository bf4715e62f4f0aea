#!/bin/bash
# register / spill / scratch table of every kernel (make asm's -Rpass-analysis output), one line per kernel
cd "$(dirname "$0")/../tilecoderaytracer_amd/csrc"
make asm 2>&1 | python3 -c "
import re,sys
cur=None; rows=[]
for line in sys.stdin:
    m=re.search(r'remark:\s+Function Name: (\S+)',line)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    m=re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)',line)
    if m and cur is not None: cur[m.group(1).strip()]=m.group(2)
print('%-34s %5s %5s %8s %8s %8s %5s'%('kernel','VGPR','SGPR','sgprSpill','vgprSpill','scratch','occ'))
for r in rows:
    print('%-34s %5s %5s %8s %8s %8s %5s'%(r['name'],r.get('VGPRs'),r.get('TotalSGPRs'),r.get('SGPRs Spill'),r.get('VGPRs Spill'),r.get('ScratchSize'),r.get('Occupancy')))
"
grep -E "codeLenInByte" rt_kernel.s | head -20
