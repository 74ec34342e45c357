#!/usr/bin/env python
"""Random read sets and random command lines against the compiled reference binary (needs oracle/_ref):
    python tools/fuzz_overlap.py plain SEED N    oracle `--step 1` (k, w, -r -g -n -m -f --dvt --minlen --maxhan --seed)
    python tools/fuzz_overlap.py step2 SEED N    oracle `--step 2` (--mode 0 / 1 / 2, --minlen --maxhan --minide --minmatch --kn --wn --cn)
    python tools/fuzz_overlap.py cli SEED N      the DEVICE command line (python -m nextdenovo_amd.minimap2_nd: --step 1 with the options above, -c,
                                                 --mode 3) -- on a GPU box, or with NDGPU_SIMT=1 under the kernel interpreter
    python tools/fuzz_overlap.py sort SEED N     the sort oracle (and, on a GPU box with NDGPU_FUZZ_DEVICE=1 or under NDGPU_SIMT=1, the device sort)
                                                 against the compiled `ovl_sort` (-k -l -H)
    python tools/fuzz_overlap.py dump SEED N     the device `seq_dump` command against the compiled one (GPU box or NDGPU_SIMT=1)
    python tools/fuzz_overlap.py mode3 SEED N    the device command line with --mode 3;   cli2: with --step 2 (GPU box or NDGPU_SIMT=1)
Round 3: 75 + 30 + 32 + 38 + 52 + 24 + 20 cases, all byte-identical.  (`-c` has a fuzzer of its own: tools/fuzz_cigar.py.)"""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mm_util as M  # noqa: E402
import fuzz_cigar as F  # noqa: E402
from nextdenovo_amd import synth  # noqa: E402

def fuzz_plain(seed, n_cases, lib):
    rng = np.random.default_rng(seed)
    n = n_cases
    bad = 0
    for it in range(n):
        seqs = F.reads(rng, 15000, 50000, 8, 25)
        wd=tempfile.mkdtemp(prefix="fp")
        seed, part = M.dump_reads(wd, [synth.codes_to_ascii(s) for s in seqs], seed_cutoff=int(rng.choice([3000,5000])))
        preset=str(rng.choice(["ava-ont","ava-pb"])); dual=bool(rng.random()<0.5)
        extra=[]; kw={}; mo=dict(mid_occ_frac=2e-4, mid_occ=0)
        if rng.random()<0.4:
            k=int(rng.choice([11,13,17,19,21])); extra+=["-k",str(k)]; kw["k"]=k
        if rng.random()<0.4:
            w=int(rng.choice([3,7,10])); extra+=["-w",str(w)]; kw["w"]=w
        if rng.random()<0.3:
            bw=int(rng.choice([100,300,1000,5000])); extra+=["-r",str(bw)]; kw["bw"]=bw
        if rng.random()<0.3:
            gg=int(rng.choice([1000,3000,20000])); extra+=["-g",str(gg)]; kw["max_gap"]=gg
        if rng.random()<0.3:
            nn=int(rng.choice([2,4,8])); extra+=["-n",str(nn)]; kw["min_cnt"]=nn
        if rng.random()<0.3:
            m=int(rng.choice([30,60,300])); extra+=["-m",str(m)]; kw["min_sc"]=m
        if rng.random()<0.3:
            f=[("0.001",dict(mid_occ_frac=1e-3)),("0.01",dict(mid_occ_frac=1e-2)),("20",dict(mid_occ=20)),("200",dict(mid_occ=200))][int(rng.integers(0,4))]; extra+=["-f",f[0]]; mo.update(f[1])
        if rng.random()<0.2:
            extra+=["--dvt"]; kw["dvt"]=1
        if rng.random()<0.2:
            ml=int(rng.choice([200,1000,3000])); extra+=["--minlen",str(ml)]; kw["minlen"]=ml
        if rng.random()<0.15:
            extra+=["--maxhan1","2000","--maxhan2","200"]; kw["maxhan1"]=2000; kw["maxhan2"]=200
        if rng.random()<0.15:
            sd=int(rng.choice([1,7,99])); extra+=["--seed",str(sd)]; kw["seed"]=sd
        t,q=seed,(part if dual and part else seed)
        try: want=M.ref_step1(t,q,os.path.join(wd,"ref.ovl"),preset,dual,tuple(extra),threads=4)
        except subprocess.CalledProcessError: print(it,"reference failed",preset,extra,flush=True); continue
        got,_=M.step1(lib, M.preset(preset,dual,**kw), M.load_set(t), M.load_set(q), **mo)
        ok=got==want; bad+=(not ok)
        print(it,"equal" if ok else "DIFFER",preset,dual,extra,len(got),len(want),"" if ok else wd,flush=True)
    print("mismatches", bad)
    return bad



def fuzz_step2(seed, n_cases, lib):
    rng = np.random.default_rng(seed)
    n = n_cases
    bad = 0
    for it in range(n):
        G=int(rng.integers(20000,70000)); depth=float(rng.uniform(12,60))
        g=synth.make_genome(G, seed=int(rng.integers(1,10**6)), n_repeats=int(rng.integers(0,4)), repeat_len=int(rng.integers(500,2000)))
        rs=synth.simulate_reads(g, depth, "hifi", seed=int(rng.integers(1,10**6)), mu=float(rng.uniform(8.3,8.9)), sigma=0.35, min_len=2000)
        seqs=list(rs.seqs)
        for t in range(int(rng.integers(0,30))):
            a=int(rng.integers(0,len(seqs)))
            if seqs[a].size>3000:
                L=int(rng.integers(2100,2900)); s0=int(rng.integers(0,seqs[a].size-L)); seqs.append(seqs[a][s0:s0+L].copy())
        wd=tempfile.mkdtemp(prefix="f2"); half=len(seqs)//2; files=[]; sets=[]
        for tag,lo,hi in (("a",0,half),("b",half,len(seqs))):
            p=os.path.join(wd,tag+".fasta")
            with open(p,"w") as f:
                for i in range(lo,hi): f.write(">%d %d 0.99\n%s\n"%(i+1,seqs[i].size,synth.codes_to_ascii(seqs[i]).decode()))
            files.append(p)
            ids=np.arange(lo+1,hi+1,dtype=np.uint32); lens=np.asarray([seqs[i].size for i in range(lo,hi)],dtype=np.uint32)
            off=np.zeros(hi-lo,dtype=np.uint64); off[1:]=np.cumsum(lens.astype(np.uint64))[:-1]
            sets.append((ids,lens,np.concatenate([seqs[i] for i in range(lo,hi)]).astype(np.uint8),off))
        preset=str(rng.choice(["ava-ont","ava-pb"])); mode=int(rng.choice([0,1,2,2]))
        extra=["-k","17","-w",str(int(rng.choice([10,17])))]; kw={"k":17,"w":int(extra[3])}
        s2=dict(minide=0.05,minmatch=100,kn=17,wn=10,cn=50 if mode==1 else 20)
        if rng.random()<0.5:
            ml=int(rng.choice([700,1000,2000,3000])); extra+=["--minlen",str(ml)]; kw["minlen"]=ml
        else: kw["minlen"]=2000
        if rng.random()<0.4:
            h1=int(rng.choice([1500,2000,8000])); extra+=["--maxhan1",str(h1)]; kw["maxhan1"]=h1
        if rng.random()<0.3:
            h2=int(rng.choice([100,300,900])); extra+=["--maxhan2",str(h2)]; kw["maxhan2"]=h2
        if rng.random()<0.3:
            mi=float(rng.choice([0.02,0.1,0.3])); extra+=["--minide",str(mi)]; s2["minide"]=mi
        if rng.random()<0.3:
            mm=int(rng.choice([50,200,500])); extra+=["--minmatch",str(mm)]; s2["minmatch"]=mm
        if mode and rng.random()<0.3:
            kn=int(rng.choice([15,17,19])); extra+=["--kn",str(kn)]; s2["kn"]=kn
        if mode and rng.random()<0.3:
            wn=int(rng.choice([5,10,15])); extra+=["--wn",str(wn)]; s2["wn"]=wn
        if mode and rng.random()<0.3:
            cn=int(rng.choice([5,20,40])); extra+=["--cn",str(cn)]; s2["cn"]=cn
        if mode==1: s2["minide"]=max(s2["minide"],0.01)
        if mode==1 and s2["cn"]==20: s2["cn"]=50   # main.c:457 cannot tell `--cn 20` from the default and makes both 50 in --mode 1
        out=os.path.join(wd,"o.ovl")
        cmd=[os.path.join(M.REFDIR,"minimap2-nd"),"--step","2",*(("--mode",str(mode)) if mode!=2 else ()),"--dual=yes","-t","3","-x",preset,*extra,files[0],files[1],files[0],"-o",out]
        try: subprocess.run(cmd,check=True,stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL)
        except subprocess.CalledProcessError: print(it,"reference failed",mode,preset,extra,flush=True); continue
        want,want_bl=open(out,"rb").read(),open(out+".bl").read()
        t0=time.time()
        got,got_bl=M.step2(lib,M.preset(preset,True,**kw),sets[0],[sets[1],sets[0]],mode,**s2)
        un=lib.nd_mm_step2_unrestated()
        ok=got==want and got_bl==want_bl; bad+=(not ok)
        print(it,"equal" if ok else "DIFFER","mode",mode,preset,extra,len(got),len(want),"unrestated",un,"%.0fs"%(time.time()-t0),"" if ok else wd,flush=True)
    print("mismatches", bad)
    return bad



def fuzz_cli(seed, n_cases, lib):
    from nextdenovo_amd import minimap2_nd
    rng = np.random.default_rng(seed)
    n = n_cases
    bad = 0
    for it in range(n):
        seqs = F.reads(rng, 12000, 25000, 8, 14)
        wd=tempfile.mkdtemp(prefix="fc")
        seed, part = M.dump_reads(wd, [synth.codes_to_ascii(s) for s in seqs], seed_cutoff=3000)
        preset=str(rng.choice(["ava-ont","ava-pb"])); dual=bool(rng.random()<0.5)
        extra=[]
        if rng.random()<0.4: extra+=["-k",str(int(rng.choice([11,13,17,19,21])))]
        if rng.random()<0.4: extra+=["-w",str(int(rng.choice([3,7,10])))]
        if rng.random()<0.3: extra+=["-r",str(int(rng.choice([100,300,1000,5000])))]
        if rng.random()<0.3: extra+=["-g",str(int(rng.choice([1000,3000,20000])))]
        if rng.random()<0.3: extra+=["-n",str(int(rng.choice([2,4,8])))]
        if rng.random()<0.3: extra+=["-m",str(int(rng.choice([30,60,300])))]
        if rng.random()<0.3: extra+=["-f",str(rng.choice(["0.001","0.01","20","200"]))]
        if rng.random()<0.2: extra+=["--dvt"]
        if rng.random()<0.2: extra+=["--minlen",str(int(rng.choice([200,1000,3000])))]
        if rng.random()<0.15: extra+=["--seed",str(int(rng.choice([1,7,99])))]
        if rng.random()<0.25: extra+=["-c"]
        if rng.random()<0.15 and "-c" not in extra: extra+=["--mode","3"]
        t,q=seed,(part if dual and part else seed)
        argv=["--step","1"]+(["--dual=yes"] if dual else [])+["-t","4","-x",preset,*extra,t,q,"-o",os.path.join(wd,"dev.ovl")]
        try: want=M.ref_step1(t,q,os.path.join(wd,"ref.ovl"),preset,dual,tuple(extra),threads=4)
        except subprocess.CalledProcessError: print(it,"reference failed",preset,extra,flush=True); continue
        t0=time.time()
        try:
            rc=minimap2_nd.run(argv); got=open(os.path.join(wd,"dev.ovl"),"rb").read()
        except SystemExit as e:
            print(it,"device refused",preset,extra,e,flush=True); continue
        ok=got==want; bad+=(not ok)
        print(it,"equal" if ok else "DIFFER",preset,dual,extra,len(got),len(want),"%.0fs"%(time.time()-t0),"" if ok else wd,flush=True)
    print("mismatches", bad)
    return bad



def fuzz_sort(seed, n_cases, lib):
    """`ovl_sort` (-k -l -H on ONT / HiFi read sets with glued chimeric reads): the sort oracle -- and with DEV the device sort -- against the
    compiled reference program."""
    import os_util as O
    from nextdenovo_amd import ovl
    olib = O.bind(lib)
    DEV = bool(os.environ.get("NDGPU_SIMT") or os.environ.get("NDGPU_FUZZ_DEVICE"))
    if DEV:
        from nextdenovo_amd import overlap
    rng = np.random.default_rng(seed)
    n = n_cases
    bad = 0
    for it in range(n):
        G=int(rng.integers(30000,90000)); depth=float(rng.uniform(15,70)); prof=str(rng.choice(["ont","ont","hifi"]))
        g=synth.make_genome(G, seed=int(rng.integers(1,10**6)), n_repeats=int(rng.integers(0,6)), repeat_len=int(rng.integers(800,3000)))
        kw=dict(mu=9.0,sigma=0.3,min_len=3000) if prof=="hifi" else dict(mu=float(rng.uniform(8.6,9.4)),sigma=0.5,min_len=1000)
        rs=synth.simulate_reads(g, depth, prof, seed=int(rng.integers(1,10**6)), **kw)
        seqs=list(rs.seqs)
        for t in range(int(rng.integers(0,25))):
            a,b=rng.integers(0,len(seqs),2); y=synth.revcomp_codes(seqs[b]) if t%2 else seqs[b]
            seqs.append(np.concatenate([seqs[a][:max(1500,seqs[a].size//2)], y[:max(1500,y.size//2)]]))
        wd=tempfile.mkdtemp(prefix="fs")
        seed,part=M.dump_reads(wd,[synth.codes_to_ascii(s) for s in seqs],seed_cutoff=int(rng.choice([4000,7000,10000])))
        preset="ava-hifi" if prof=="hifi" else "ava-ont"
        files=[]
        try:
            if part:
                M.ref_step1(seed,part,os.path.join(wd,"a.ovl"),preset,True); files.append(os.path.join(wd,"a.ovl"))
            M.ref_step1(seed,seed,os.path.join(wd,"b.ovl"),preset,False); files.append(os.path.join(wd,"b.ovl"))
        except subprocess.CalledProcessError: print(it,"step1 failed",flush=True); continue
        idx=os.path.join(wd,"db",".input.seed.001.idx")
        k=int(rng.choice([6,12,20,30,40,60])); flank=int(rng.choice([100,300,300,1000])); hq=bool(prof=="hifi" and rng.random()<0.7) or bool(rng.random()<0.15)
        try: want,want_bl=O.ref_sort(wd,idx,files,k=k,flank=flank if flank!=300 else None,hq=hq)
        except Exception as e: print(it,"ref sort failed",e,flush=True); continue
        sl,mn=O.read_idx(idx)
        blob,bl,_=O.oracle_sort(olib,[ovl.decode_ovl(f) for f in files],sl,mn,max_bin_cov=k,flank=flank,hq=hq)
        ok=blob==want and bl==want_bl
        if DEV:
            recs,dbl,_=overlap.sort_overlaps([overlap.from_decoded(ovl.decode_ovl(f)) for f in files], sl, mn, k, flank, hq=hq)
            ok = ok and overlap.encode(recs,np.zeros(2,dtype=np.uint32))==want and ''.join('%d %s\n'%x for x in dbl)==want_bl
        bad+=(not ok)
        print(it,"equal" if ok else "DIFFER",prof,"k",k,"l",flank,"H",hq,len(blob),len(want),"" if ok else wd,flush=True)
    print("mismatches", bad)
    return bad



def fuzz_dump(seed, n_cases, lib):
    """`seq_dump` (messy FASTA / FASTQ[.gz]: lower case, N and IUPAC codes, CRLF, empty records, multi-file fofn; -f -s -b -n): the device
    command (the 2-bit packer kernel) against the compiled reference program, every output file."""
    REF = os.path.join(ROOT, 'oracle', '_ref', 'seq_dump')
    rng = np.random.default_rng(seed)
    n = n_cases
    bad = 0
    import gzip
    from nextdenovo_amd import seq_dump

    def seq(k):
        alpha=list("ACGT") if rng.random()<0.6 else list("ACGTacgtNnRYKMU-")
        p=np.ones(len(alpha)); p[:4]=20; p/=p.sum()
        return "".join(rng.choice(alpha,k,p=p))
    for it in range(n):
        d=tempfile.mkdtemp(prefix="fd"); paths=[]
        for fi in range(int(rng.integers(1,5))):
            fq=rng.random()<0.4; gz=rng.random()<0.4
            path=os.path.join(d,"f%d.%s%s"%(fi,"fq" if fq else "fa",".gz" if gz else ""))
            with (gzip.open(path,"wt",newline="") if gz else open(path,"w",newline="")) as f:
                if rng.random()<0.3 and not fq: f.write("leading junk\n")
                for r in range(int(rng.integers(1,40))):
                    L=int(rng.choice([0,1,15,16,17,31,32,33,int(rng.integers(1,400)),int(rng.integers(400,6000)),int(rng.integers(400,6000))]))
                    s=seq(L); nl="\r\n" if rng.random()<0.2 else "\n"
                    if fq:
                        q="".join(rng.choice(list("@+>IJK#5!~"),L))
                        f.write("@q%d_%d comment%s%s%s+%s%s%s"%(fi,r,nl,s,nl,"" if rng.random()<0.5 else "q%d"%r,nl,q+nl))
                    else:
                        w=int(rng.choice([50,60,70,80,10**7])); f.write(">r%d_%d%s"%(fi,r," c" if rng.random()<0.5 else "")+nl)
                        for k in range(0,max(L,1),w): f.write(s[k:k+w]+nl)
                        if rng.random()<0.1: f.write(nl)
            paths.append(path)
        fofn=os.path.join(d,"in.fofn"); open(fofn,"w").write("\n".join(paths)+"\n")
        argv=["-f",str(int(rng.choice([1,16,100,500,1000]))),"-s",str(int(rng.choice([500,1001,2500,4000]))),"-b",str(rng.choice(["0","3k","20k","1g"])),"-n",str(int(rng.integers(1,4)))]
        ref,mine=os.path.join(d,"ref"),os.path.join(d,"mine")
        r=subprocess.run([REF,*argv,"-d",ref,fofn],stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL)
        if r.returncode!=0: print(it,"reference failed",argv,flush=True); continue
        try: rc=seq_dump.run([*argv,"-d",mine,fofn])
        except SystemExit as e: print(it,"device refused",argv,e,flush=True); continue
        names=sorted(os.listdir(ref)); ok=names==sorted(os.listdir(mine))
        diff=[]
        if ok:
            for nme in names:
                if open(os.path.join(ref,nme),"rb").read()!=open(os.path.join(mine,nme),"rb").read(): ok=False; diff.append(nme)
        bad+=(not ok)
        print(it,"equal" if ok else "DIFFER",argv,len(names),"files",diff,"" if ok else d,flush=True)
    print("mismatches", bad)
    return bad



def fuzz_mode3(seed, n_cases, lib):
    """`--step 1 --mode 3` (HiFi / ONT / CLR reads; --dvt --df -f --maxhan --minlen): the device command line against the reference binary."""
    from nextdenovo_amd import minimap2_nd
    rng = np.random.default_rng(seed)
    n = n_cases
    bad = 0
    for it in range(n):
        prof=str(rng.choice(["hifi","hifi","ont"]))
        g=synth.make_genome(int(rng.integers(30000,80000)), seed=int(rng.integers(1,10**6)), n_repeats=int(rng.integers(0,5)), repeat_len=int(rng.integers(800,3000)))
        kw=dict(mu=float(rng.uniform(8.7,9.2)),sigma=0.3,min_len=3000) if prof=="hifi" else dict(mu=8.8,sigma=0.5,min_len=1500)
        rs=synth.simulate_reads(g, float(rng.uniform(12,35)), prof, seed=int(rng.integers(1,10**6)), **kw)
        wd=tempfile.mkdtemp(prefix="f3")
        seed,part=M.dump_reads(wd,[synth.codes_to_ascii(s) for s in rs.seqs],seed_cutoff=int(rng.choice([5000,8000])))
        preset="ava-hifi" if prof=="hifi" else str(rng.choice(["ava-ont","ava-pb"])); dual=bool(rng.random()<0.5)
        extra=["--mode","3"]
        if rng.random()<0.4: extra+=["--dvt"]
        if rng.random()<0.3: extra+=["--df",str(rng.choice(["0.05","0.2","0.4"]))]
        if rng.random()<0.3 and prof=="hifi": extra+=["-f",str(rng.choice(["40","200","0.001"]))]
        if rng.random()<0.3: extra+=["--maxhan1",str(int(rng.choice([1000,3000]))),"--maxhan2",str(int(rng.choice([100,500])))]
        if rng.random()<0.3: extra+=["--minlen",str(int(rng.choice([300,2000])))]
        t,q=seed,(part if dual and part else seed)
        try: want=M.ref_step1(t,q,os.path.join(wd,"ref.ovl"),preset,dual,tuple(extra),threads=4)
        except subprocess.CalledProcessError: print(it,"reference failed",preset,extra,flush=True); continue
        argv=["--step","1"]+(["--dual=yes"] if dual else [])+["-t","4","-x",preset,*extra,t,q,"-o",os.path.join(wd,"dev.ovl")]
        t0=time.time()
        try: minimap2_nd.run(argv); got=open(os.path.join(wd,"dev.ovl"),"rb").read()
        except SystemExit as e: print(it,"device refused",extra,e,flush=True); continue
        ok=got==want; bad+=(not ok)
        print(it,"equal" if ok else "DIFFER",preset,dual,extra,len(got),len(want),"%.0fs"%(time.time()-t0),"" if ok else wd,flush=True)
    print("mismatches", bad)
    return bad



def fuzz_cli2(seed, n_cases, lib):
    """`--step 2` (modes 0 / 1 / 2 and their options): the device command line against the reference binary (`.ovl` and `.bl`)."""
    from nextdenovo_amd import minimap2_nd
    rng = np.random.default_rng(seed)
    n = n_cases
    bad = 0
    for it in range(n):
        G=int(rng.integers(15000,35000)); depth=float(rng.uniform(10,40))
        g=synth.make_genome(G, seed=int(rng.integers(1,10**6)), n_repeats=int(rng.integers(0,4)), repeat_len=int(rng.integers(500,2000)))
        rs=synth.simulate_reads(g, depth, "hifi", seed=int(rng.integers(1,10**6)), mu=float(rng.uniform(8.3,8.9)), sigma=0.35, min_len=2000)
        seqs=list(rs.seqs)
        for t in range(int(rng.integers(0,30))):
            a=int(rng.integers(0,len(seqs)))
            if seqs[a].size>3000:
                L=int(rng.integers(2100,2900)); s0=int(rng.integers(0,seqs[a].size-L)); seqs.append(seqs[a][s0:s0+L].copy())
        wd=tempfile.mkdtemp(prefix="f2"); half=len(seqs)//2; files=[]; sets=[]
        for tag,lo,hi in (("a",0,half),("b",half,len(seqs))):
            p=os.path.join(wd,tag+".fasta")
            with open(p,"w") as f:
                for i in range(lo,hi): f.write(">%d %d 0.99\n%s\n"%(i+1,seqs[i].size,synth.codes_to_ascii(seqs[i]).decode()))
            files.append(p)
            ids=np.arange(lo+1,hi+1,dtype=np.uint32); lens=np.asarray([seqs[i].size for i in range(lo,hi)],dtype=np.uint32)
            off=np.zeros(hi-lo,dtype=np.uint64); off[1:]=np.cumsum(lens.astype(np.uint64))[:-1]
            sets.append((ids,lens,np.concatenate([seqs[i] for i in range(lo,hi)]).astype(np.uint8),off))
        preset=str(rng.choice(["ava-ont","ava-pb"])); mode=int(rng.choice([0,1,2,2]))
        extra=["-k","17","-w",str(int(rng.choice([10,17])))]; kw={"k":17,"w":int(extra[3])}
        s2=dict(minide=0.05,minmatch=100,kn=17,wn=10,cn=50 if mode==1 else 20)
        if rng.random()<0.5:
            ml=int(rng.choice([700,1000,2000,3000])); extra+=["--minlen",str(ml)]; kw["minlen"]=ml
        else: kw["minlen"]=2000
        if rng.random()<0.4:
            h1=int(rng.choice([1500,2000,8000])); extra+=["--maxhan1",str(h1)]; kw["maxhan1"]=h1
        if rng.random()<0.3:
            h2=int(rng.choice([100,300,900])); extra+=["--maxhan2",str(h2)]; kw["maxhan2"]=h2
        if rng.random()<0.3:
            mi=float(rng.choice([0.02,0.1,0.3])); extra+=["--minide",str(mi)]; s2["minide"]=mi
        if rng.random()<0.3:
            mm=int(rng.choice([50,200,500])); extra+=["--minmatch",str(mm)]; s2["minmatch"]=mm
        if mode and rng.random()<0.3:
            kn=int(rng.choice([15,17,19])); extra+=["--kn",str(kn)]; s2["kn"]=kn
        if mode and rng.random()<0.3:
            wn=int(rng.choice([5,10,15])); extra+=["--wn",str(wn)]; s2["wn"]=wn
        if mode and rng.random()<0.3:
            cn=int(rng.choice([5,20,40])); extra+=["--cn",str(cn)]; s2["cn"]=cn
        if mode==1: s2["minide"]=max(s2["minide"],0.01)
        if mode==1 and s2["cn"]==20: s2["cn"]=50   # main.c:457 cannot tell `--cn 20` from the default and makes both 50 in --mode 1
        out=os.path.join(wd,"o.ovl")
        cmd=[os.path.join(M.REFDIR,"minimap2-nd"),"--step","2",*(("--mode",str(mode)) if mode!=2 else ()),"--dual=yes","-t","3","-x",preset,*extra,files[0],files[1],files[0],"-o",out]
        try: subprocess.run(cmd,check=True,stdout=subprocess.DEVNULL,stderr=subprocess.DEVNULL)
        except subprocess.CalledProcessError: print(it,"reference failed",mode,preset,extra,flush=True); continue
        want,want_bl=open(out,"rb").read(),open(out+".bl").read()
        t0=time.time()
        dout=os.path.join(wd,"dev.ovl")
        dargv=["--step","2",*(("--mode",str(mode)) if mode!=2 else ()),"--dual=yes","-t","3","-x",preset,*extra,files[0],files[1],files[0],"-o",dout]
        try: minimap2_nd.run(dargv)
        except SystemExit as e: print(it,"device refused",extra,e,flush=True); continue
        got,got_bl=open(dout,"rb").read(),open(dout+".bl").read()
        un=0
        ok=got==want and got_bl==want_bl; bad+=(not ok)
        print(it,"equal" if ok else "DIFFER","mode",mode,preset,extra,len(got),len(want),"unrestated",un,"%.0fs"%(time.time()-t0),"" if ok else wd,flush=True)
    print("mismatches", bad)
    return bad



def main():
    mode, seed, n_cases = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    lib = M.bind(C.CDLL(os.path.join(ROOT, "oracle", "libndoracle.so")))
    lib.nd_mm_step2_unrestated.restype = C.c_int64
    if mode in ("cli", "sort", "dump", "mode3", "cli2") and os.environ.get("NDGPU_SIMT"):
        sys.path.insert(0, os.path.join(ROOT, "tests", "simt"))
        import build_simt
        from nextdenovo_amd import overlap
        os.environ.setdefault("NDGPU_CONTEXTS", "1")
        overlap._lib = overlap._bind(C.CDLL(build_simt.build_overlap()))
    return 1 if {"plain": fuzz_plain, "step2": fuzz_step2, "cli": fuzz_cli, "sort": fuzz_sort, "dump": fuzz_dump, "mode3": fuzz_mode3, "cli2": fuzz_cli2}[mode](seed, n_cases, lib) else 0


if __name__ == "__main__":
    sys.exit(main())
