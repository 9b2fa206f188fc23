#!/usr/bin/env python3
"""Study (build container only, needs /root/reference): which code produced the reference's published D*(x) files?

Runs the GEN-1 script's own function text (LiCoO2_kmc.py: rectangular 4 x 4 local lattices read from the real
occupancies, all 13 free sites of master_dictionary.p's key space, 1.48 eV dumbbell) in the shape its
LiCoO2_kmc.out header shows (10 x 10, 100 + 100 cycles), in dictionary mode (`dict`) or with the live cluster
expansion.  Nothing of the reference is copied: line ranges are sliced at run time, print statements dropped.
Result (profiles/r2_gen1_study.txt): dictionary mode gives D*(0.1) ~ 1e-19, D*(0.9) ~ 1e-11..4e-11 -- the decades of
data_run_4_data (7.5e-19, 6.4e-12) -- while the hex path (what this repository accelerates, bit for bit) gives
1e-14 / 5e-11 on the master fold: the published files come from the gen-1 model, whose rates read sites three
steps from the hopping atom, outside the 18-site hexagon of hex_dict.p.  TEST INFRASTRUCTURE / STUDY ONLY.
    python oracle/study_gen1_dstar.py [dict]"""
import numpy as np, math, itertools, sys, re, time, pickle
sys.path.insert(0,'/root/repo')
from kmc_amd.tables import load_table_pickle
lines=open('/root/reference/LiCoO2_kmc.py',encoding='utf-8',errors='replace').readlines()
def rng(a,b): return ''.join(lines[a-1:b])
g=dict(np=np,math=math,itertools=itertools,sys=sys,plt=None,time=time)
src=rng(23,36)+rng(38,235)+rng(241,408)+rng(424,485)
src=src.replace('xrange','range')
src=re.sub(r'^(\s*)print\b.*$', r'\1pass', src, flags=re.M)
try:
    exec(compile(src,'gen1','exec'),g)
except SyntaxError as e:
    print('syntax', e, e.lineno); raise
use_dict = len(sys.argv)>1 and sys.argv[1]=='dict'
if use_dict:
    g['master_dictionary']=load_table_pickle('/root/reference/master_dictionary.p')
    # flip use_dictionary_mode
    src2=rng(211,235).replace('use_dictionary_mode = False','use_dictionary_mode = True')
    exec(compile(src2,'ghp','exec'),g)
def run(n,x,pre,sim,seed):
    np.random.seed(seed)
    size=[n,n,1]; g['mc_lattice_size']=size; g['Li_concentration']=x
    lat=np.zeros(size)+np.float32(1); li=0
    while li<=x*n*n-1:
        i=np.random.randint(0,n); j=np.random.randint(0,n); k=np.random.randint(0,1)
        if lat[i][j][k]==1: lat[i][j][k]=np.float32(-1); li+=1
    P=np.tile(np.zeros(6),(n,n,1,1)); dist=np.tile([-1,0,0],(n,n,1,1))
    g['use_initialize_mode']=True
    P=g['kmc_evolve'](lat,P,0,0,True)
    g['use_initialize_mode']=False
    for _ in range(pre): lat,P=g['kmc_step'](lat,dist,P,True)
    for i,j,k in itertools.product(range(n),range(n),range(1)):
        dist[i][j][k]=[i,j,k] if lat[i][j][k]==-1 else [-1,0,0]
    T=0
    for _ in range(sim):
        lat,dist,t,P=g['kmc_step'](lat,dist,P,False); T+=t
    d,_,dj=g['get_diffusion_coefficient'](dist,size,T,g['lattice_constant'],x)
    return d,dj,T
for x in (0.5,0.9,0.1):
    t0=time.time()
    out=[run(10,x,100,100,s) for s in range(3)]
    print('gen1', 'dict' if use_dict else 'live', 'x',x,'D*',[o[0] for o in out],'time',[o[2] for o in out],'%.1fs'%(time.time()-t0),flush=True)
