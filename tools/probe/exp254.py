import os, sys, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
from slam_amd import lib as K, synth, shard
W,H,P,NF=1241,376,127,2000
frames = synth.render_corridor_frames(W,H,shard.rank_seed(0x5EED0002,0),P+1,workers=8)
for rep in (1,2):
    ctx=K.Context(K.default_config(width=W,height=H,num_slots=(P+1)*rep,max_points=2048))
    for r in range(rep):
        for s,f in enumerate(frames): ctx.upload_image(r*(P+1)+s,f)
    pts=np.zeros((P*rep,NF,2),np.float32)
    slots=list(range((P+1)*rep)); ctx.build_pyramids(slots)
    gp=K.gftt_params(NF,0.01,9.0,3,0,0.0)
    for i in range(P):
        (xy,_),=ctx.detect_gftt(i,[(0,0,W,H)],gp)
        md=9.0
        while len(xy)<NF:
            md*=0.9; (xy,_),=ctx.detect_gftt(i,[(0,0,W,H)],K.gftt_params(NF,0.01,md,3,0,0.0))
        for r in range(rep): pts[r*P+i]=xy
    dp=torch.from_numpy(pts).cuda(); dn=torch.zeros_like(dp); ds=torch.zeros((P*rep,NF),dtype=torch.uint8,device='cuda'); de=torch.zeros((P*rep,NF),device='cuda')
    prev=[r*(P+1)+i for r in range(rep) for i in range(P)]; nxt=[p+1 for p in prev]
    lk=K.lk_params()
    def step():
        ctx.build_pyramids(slots); ctx.pyrlk_batch_dev(prev,nxt,dp.data_ptr(),NF,dn.data_ptr(),ds.data_ptr(),de.data_ptr(),lk)
    for _ in range(3): step()
    ctx.sync(); t=time.perf_counter()
    for _ in range(20): step()
    ctx.sync(); el=(time.perf_counter()-t)/20
    print(rep, "pairs", P*rep, "ms/step", round(el*1e3,4), "M/s", round(int(ds.sum().item())/el/1e6,2))
    ctx.close()
