"""dev tool: the parallel gzip inflater (bcalm_amd/_build/pgz_cat) against zlib on random texts and deflate parameters: fuzz_pgz.py SECONDS SEED (writes f.gz in the cwd)"""
import random, zlib, gzip, subprocess, sys, time, os
EXE="/root/repo/bcalm_amd/_build/pgz_cat"
t_end=time.time()+float(sys.argv[1]); seed=int(sys.argv[2]); n=0; handled=0
while time.time()<t_end:
    seed+=1; rng=random.Random(seed)
    kind=rng.randrange(4)
    if kind==0:
        g="".join(rng.choice("ACGT") for _ in range(rng.choice([2000,50000])))
        L=rng.choice([36,100,150,300,5000]); recs=[]
        for i in range(min(rng.randrange(2000,20000), 3000000//(2*min(L,len(g)-1)+20))):
            l=min(L,len(g)-1); s=rng.randrange(0,len(g)-l); r=g[s:s+l]
            q="".join(chr(33+rng.randrange(2,41)) for _ in range(l)) if rng.random()<0.7 else "I"*l
            recs.append("@%s.%d\n%s\n+\n%s\n"%(rng.choice(["SRR","ERR1234","x"]),i,r,q))
        t="".join(recs).encode()
    elif kind==1:
        t="".join(">s%d\n%s\n"%(i,"\n".join("".join(rng.choice("ACGTN") for _ in range(60)) for _ in range(rng.randrange(1,200)))) for i in range(rng.randrange(5,300))).encode()
    elif kind==2:
        t=(("@r\n"+"".join(rng.choice("AC") for _ in range(50))+"\n+\n"+"#"*50+"\n")*rng.randrange(1000,50000)).encode()
    else:
        t="".join(">%d\r\n%s\r\n"%(i,"".join(rng.choice("acgt") for _ in range(rng.randrange(1,500)))) for i in range(rng.randrange(100,20000))).encode()
    c=zlib.compressobj(rng.choice([1,2,3,5,6,8,9]),zlib.DEFLATED,16+rng.choice([9,10,13,15]),rng.choice([1,2,3,5,8,9]),rng.choice([zlib.Z_DEFAULT_STRATEGY,zlib.Z_FILTERED,zlib.Z_RLE,zlib.Z_HUFFMAN_ONLY]))
    fe=rng.choice([0,0,0,4099,65537]); blob=b""
    if fe:
        for i in range(0,len(t),fe): blob+=c.compress(t[i:i+fe])+c.flush(rng.choice([zlib.Z_SYNC_FLUSH,zlib.Z_FULL_FLUSH,zlib.Z_NO_FLUSH]) )
        blob+=c.flush()
    else: blob=c.compress(t)+c.flush()
    if rng.random()<0.3:
        cut=len(t)//2; blob=gzip.compress(t[:cut],rng.choice([1,6,9]))+blob[:0]+gzip.compress(t[cut:],6)
    open("f.gz","wb").write(blob)
    r=subprocess.run([EXE,"f.gz",str(rng.choice([2,3,8])),str(rng.choice([1024,5000,30000,200000]))],capture_output=True,timeout=120)
    n+=1
    if r.returncode==0:
        handled+=1
        if r.stdout!=t: print("MISMATCH seed",seed); open("bad_%d.gz"%seed,"wb").write(blob); sys.exit(1)
    elif r.returncode!=2 or r.stdout!=b"": print("BAD rc",r.returncode,"seed",seed,r.stderr[:200]); open("bad_%d.gz"%seed,"wb").write(blob); sys.exit(1)
print("cases",n,"handled",handled,"all consistent")
