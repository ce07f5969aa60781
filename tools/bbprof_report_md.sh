#!/bin/bash
# usage: mk_bbprof_md.sh <bbprof dir>  -> profiles/r03_bbprof.md
D=$1
{
echo "# Round 3: basic-block profile of \`render_kernel<false>\` (\`tools/bbprof.py\`, \`tools/bbprof.sh\`)"
echo
echo "Method: the device assembly of \`libcpuvox_gpu\` is compiled once (\`-gline-tables-only\`: same code, plus \`.loc\`); for every basic block of the kernel a variant is assembled with one"
echo "\`s_mov_b32 vcc_lo, vcc_lo\` (counted by \`SQ_INSTS_SALU\`, changes nothing) at the top of that block and linked into its own library; one process renders the same 32 frames"
echo "(first 32 bench poses, 1080p, 2048^3 world) once with each library under \`rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU\`; (count of variant b) - (count of the unmodified build) = executions of block b."
echo "Executions x static instruction counts = the dynamic instruction budget.  Self-check (first lines of the raw report): the sum over blocks reproduces the measured \`SQ_INSTS_VALU\` exactly."
echo
echo '```'
head -4 $D/bbprof.txt
echo '```'
echo
echo "## Dynamic instructions per wave-step, by part of ExecuteRay"
echo
python3 tools/bbprof_sections.py $D
echo
echo "(Blocks are attributed by the highest line of \`trace_ray\`'s own body among their instructions; the four \`f3_madd\` of the Q corners (\`:289-293\`, 24 instructions per drawn column) land in the clip / run-selection rows.  Round-2 kernel, same method: 690 per step, 73 of them \`v_mov\`, ~215 scalar.)"
echo
echo "## \`s_waitcnt\` sites (executions per wave-step; source line of \`cvx_kernels.h\`)"
echo
echo "| wait | source line | per wave-step | what it waits for |"
echo "|---|---|---|---|"
python3 - $D <<'PY'
import json,re,collections,sys,os
d=sys.argv[1]
m=json.load(open(d+'/blocks.json')); p=json.load(open(d+'/profile.json'))
ex={int(k):v for k,v in p['executions'].items()}
lines=open(d+'/device.s').read().split('\n')
heads=[]
for i,b in enumerate(m):
    a=b['line']; z=m[i+1]['line'] if i+1<len(m) else len(lines)
    if sum(1 for l in lines[a:z] if 'global_load_dwordx4' in l)>=2 and ex.get(b['block'],0)>1000: heads.append((i,ex[b['block']]))
half=len(heads)//2; split=heads[half][0]-3
sA=sum(e for i,e in heads if i<split); sB=sum(e for i,e in heads if i>=split)
lo,hi,S=(0,split,sA) if sA>=sB else (split,len(m),sB)
src=open(os.path.join(os.path.dirname(os.path.abspath(d)),'..','cpuvox_amd','csrc','cvx_kernels.h')).read().split('\n') if False else None
kfile=next((mm.group(1) for l in lines for mm in [re.match(r'\s*\.file\s+(\d+)\s+.*cvx_kernels\.h"', l)] if mm), None)
rows=collections.Counter()
for i in range(lo,hi):
    b=m[i]; a=b['line']; z=m[i+1]['line'] if i+1<len(m) else len(lines)
    e=ex.get(b['block'],0); loc=None
    for l in lines[a:z]:
        s=l.strip()
        mm=re.match(r"\.loc\s+(\d+)\s+(\d+)",s)
        if mm: loc=int(mm.group(2)) if (kfile is None or mm.group(1)==kfile) else loc; continue
        if s.startswith('s_waitcnt'): rows[(s.split(None,1)[1],loc)]+=e
ksrc=open(os.path.join(os.getcwd(),'cpuvox_amd','csrc','cvx_kernels.h')).read().split('\n')
def ln(pat, after=0):
    return next(i+1 for i,l in enumerate(ksrc) if i+1>after and pat in l)
L_ld=ln('__device__ __forceinline__ uint4 ld4('); L_st=ln('void st_pixel('); L_rec=ln('uint32_t record_offset(')
L_dda=ln('bool dda_step('); L_scan=ln('int scan_up('); L_scan_end=ln('bits of word w that fall')
L_walk=ln('// ---- element loop, :424-611'); L_portion=ln('const float portionBottom = '); L_side=ln('// side of the run, :484-542')
L_sidepix=ln('// pixel loop :519-533'); L_sidepix_end=ln('CVX_END(5);'); L_facepix=ln('// :595-603'); L_facepix_end=ln('CVX_END(7);')
L_step=ln('auto columnStep = '); L_step_end=ln('while (go) {')
def what(w,loc):
    if loc is None or loc==0: return "compiler-placed (join)"
    if L_ld<=loc<=L_ld+2 or L_rec-1<=loc<=L_rec+5 or L_step<=loc<=L_step_end or L_dda<=loc<=L_dda+20: return "record of the column now being processed (its two loads were issued one step earlier as the look-ahead)" if 'vmcnt' in w else ""
    if L_scan<=loc<=L_scan_end+30: return "LDS mask word of a horizon scan"
    if L_sidepix<=loc<=L_sidepix_end and 'lgkm' in w: return "LDS mask word of the side pixel loop"
    if L_facepix<=loc<=L_facepix_end and 'lgkm' in w: return "LDS mask word of the face pixel loop"
    if L_st-1<=loc<=L_st+3: return "colour loads ahead of the pixel stores"
    if L_walk<=loc<L_portion: return "record words of the current column / run-list entry (columns with > 2 solid runs)"
    if L_portion<=loc<=L_side+40: return "face colour / record words before the side projection"
    return ""
for (w,loc),e in sorted(rows.items(), key=lambda kv:-kv[1]):
    if e/S<0.01: continue
    print(f"| `{w}` | {loc} | {e/S:.3f} | {what(w,loc)} |")
print()
print(f"Total: {sum(rows.values())/S:.2f} `s_waitcnt` per wave-step.")
PY
echo
echo "## The 40 heaviest blocks (raw report: block, asm line, source lines, static valu / salu / branch / memory, executions, dynamic instructions)"
echo
echo '```'
sed -n 5,45p $D/bbprof.txt
echo '```'
} > profiles/r03_bbprof.md
