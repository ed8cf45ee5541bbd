import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import oracle as O
from tests.util import make_experts, rand_bf16, upload
from krasis_amd import GpuPrefillManager, KrasisEngine, ModelConfig
def run(H,I,E,k,M,n_shared,hot):
    rng=np.random.default_rng(0)
    experts=make_experts(rng,E,H,I); shared=make_experts(rng,1,H,n_shared*I)[0] if n_shared else None
    eng=KrasisEngine(); eng.configure(ModelConfig(H,I,E,k,1,n_shared,1.0)); upload(eng,0,experts,shared)
    mgr=GpuPrefillManager(eng,k)
    x=rand_bf16(rng,(M,H)); ids=np.stack([rng.choice(E,k,replace=False) for _ in range(M)]).astype(np.int32)
    if hot: ids[:,0]=5
    w=rng.random((M,k)).astype(np.float32)
    xt=torch.from_numpy(x.view(np.int16)).cuda().view(torch.bfloat16)
    out=mgr.forward(0,xt,torch.from_numpy(ids).cuda(),torch.from_numpy(w).cuda())
    torch.cuda.synchronize()
    got=out.view(torch.int16).cpu().numpy().view(np.uint16)
    out2=np.empty((M,H),np.uint16); eng.forward_moe_direct(0,x.ctypes.data,ids.ctypes.data,w.ctypes.data,out2.ctypes.data,M,k)
    bad=np.where((got!=out2).any(1))[0]
    print((H,I,E,k,M,n_shared,hot),"mismatching tokens:",len(bad),bad[:10], "zeros rows:", int((got==0).all(1).sum()))
run(256,128,8,2,200,0,False)
run(512,384,16,4,333,0,False)
run(512,384,16,4,333,0,True)
run(512,256,16,4,333,0,True)
run(512,384,16,4,333,1,False)
run(512,128,16,4,100,1,False)
