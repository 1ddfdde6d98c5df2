#!/usr/bin/env python3
"""How well do the dispatcher's fitted cost tables (gemm.cpp TILE_COSTS / TILE_COSTS_ROW_MAJOR_B) generalise?  No GPU needed.

Inputs: two seeded random audits that print EVERY forced kernel's time (AUDIT_ALL_TIMES=1 tools/dev/random_audit.py), committed as
profiles/r06_audit_times_fit.txt (seeds 701-708) and profiles/r06_audit_times_held_out.txt (seeds 801-804 and 601-604): 3 072 cases, of which
~670 per file lie in the tables' domain (both extents past 128, K >= 512, at most three rounds of the square tile).
For each rhs layout it prints the regret (time of the table's choice / best measured tile kernel) of
  * the table as committed (fitted in round 5 on hand-picked A/B sweeps), on both seed sets;
  * form A: the same model refitted on the fit seeds, judged on the held-out seeds;
  * form B: the chip-pace term per ROUND (rounds x CUs x tile FLOPs / P) instead of per padded tile, refitted the same way.
Outcome (profiles/r06_tile_table_holdout.txt): the committed tables are at 0.16 % mean regret on the held-out seeds with 1-2 shapes of 330 more than
10 % behind; neither refit improves on them (form B is worse: the 128 x 128 kernel's two co-resident workgroups do not fit it).  What the audits flag
lies outside the tables' domain (few rows / columns, K <= 256, one thin side) -- the named thresholds of profiles/dispatch_rules.md.
usage: python tools/dev/tile_table_holdout.py"""
import os, re, sys
import numpy as np
os.chdir(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from scipy.optimize import least_squares
TILES={"lp128":(128,128),"lp256x128":(256,128),"lp256w4":(256,256),"lp256x192":(256,192),"lp192x192":(192,192)}
CUS,F0=256,0.7
def parse(path):
    out={0:{},1:{}}; nn=0
    for line in open(path):
        if line.startswith("== rhs"): nn = 1 if "row-major" in line else 0; continue
        m=re.match(r"\s*(\d+)x\s*(\d+)x\s*(\d+): AUTO -> (\S+)\s+([\d.]+) us.*\| (.*)",line)
        if not m: continue
        s=tuple(int(m.group(i)) for i in (1,2,3))
        d={k:float(t) for k,t in re.findall(r"(\w+) ([\d.]+)",m.group(6))}
        d["_auto"]=(m.group(4),float(m.group(5)))
        out[nn][s]=d
    return out
def indomain(s):
    m,n,k=s; t256=-(-m//256)*-(-n//256)
    return min(m,n)>128 and k>=512 and t256<=768
def rounds_of(t):
    w,l=divmod(t,CUS); return w+(F0+(1-F0)*l/CUS if l else 0.0)
def predict(form,kern,s,c,f,p):
    m,n,k=s; tm,tn=TILES[kern]; tiles=-(-m//tm)*-(-n//tn); nk=k//64; r=rounds_of(tiles)
    own=r*(nk*c+f)
    if form=="A": chip=2.0*tiles*tm*tn*k/(p*1e6)
    else: chip=r*CUS*2.0*tm*tn*k/(p*1e6)
    return max(own,chip)
def fit(form,data):
    params={}
    for kern in TILES:
        pts=[(s,d[kern]) for s,d in data.items() if kern in d and indomain(s)]
        def resid(x): return [np.log(predict(form,kern,s,*x)/t) for s,t in pts]
        best=None
        for p0 in (800.,1100.,1400.):
            r=least_squares(resid,x0=[0.8,6.0,p0],bounds=([0.05,0,300],[5,40,2600]))
            if best is None or r.cost<best.cost: best=r
        params[kern]=best.x
        err=np.abs(np.exp(resid(best.x))-1)
        print(f"   {form} {kern:10s} c={best.x[0]:.3f} f={best.x[1]:5.2f} P={best.x[2]:6.0f}  n={len(pts)} med|err| {100*np.median(err):.1f}% max {100*err.max():.1f}%")
    return params
def regret(form,params,data,label):
    rs=[];w=[]
    for s,d in data.items():
        if not indomain(s): continue
        cand=[k for k in TILES if k in d]
        if len(cand)<2: continue
        ch=min(cand,key=lambda k:predict(form,k,s,*params[k])); be=min(cand,key=lambda k:d[k])
        r=d[ch]/d[be]; rs.append(r)
        if r>1.10 and d[ch]-d[be]>2: w.append((s,ch,be,round(r,3)))
    rs=np.array(rs)
    print(f"{label}: n={len(rs)} mean regret {100*(rs.mean()-1):.2f}% >5%: {(rs>1.05).sum()} >10%: {(rs>1.10).sum()} worst {rs.max():.2f}")
    return w
CUR={0:{"lp128":(0.558,1.84,890.),"lp256x128":(0.873,6.42,1167.),"lp256w4":(1.313,7.46,1351.),"lp256x192":(1.043,7.12,1307.),"lp192x192":(0.829,6.41,1219.)},
     1:{"lp128":(0.623,1.53,790.),"lp256x128":(0.945,6.95,1010.),"lp256w4":(1.237,9.45,1256.),"lp256x192":(1.014,7.64,1262.),"lp192x192":(0.827,5.59,1209.)}}
F=parse("profiles/r06_audit_times_fit.txt"); H=parse("profiles/r06_audit_times_held_out.txt")
for nn in (0,1):
    print("==== layout", "row-major" if nn else "[N][K]")
    regret("A",CUR[nn],F[nn],"current table on fit seeds"); w0=regret("A",CUR[nn],H[nn],"current table on held-out")
    for form in ("A","B"):
        p=fit(form,F[nn]); regret(form,p,F[nn],f"form {form} refit, fit seeds"); w=regret(form,p,H[nn],f"form {form} refit, held-out")
        for x in w: print("      ",x)
    print("  current misses held-out:"); [print("      ",x) for x in w0]
