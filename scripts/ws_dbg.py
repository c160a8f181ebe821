import sys, os, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "scripts"))
from lora_amd import _C
from kbench import timeit
DEV="cuda:0"; r=4
for (M,K,N) in ((16384,320,320),(16384,320,2560),(4096,640,640)):
    x=torch.randn(M,K,device=DEV).to(torch.bfloat16); W=(torch.randn(N,K,device=DEV)*0.03).to(torch.bfloat16)
    A=torch.randn(r,K,device=DEV)*0.25; B=torch.randn(N,r,device=DEV)*0.05
    y=torch.empty(M,N,dtype=torch.bfloat16,device=DEV)
    site=dict(wp=_C.ws_pack(W),N=N,down=A,up=B,scale=1e-3,y=y)
    med,_=timeit(lambda:_C.linear_ws(x,[site]),10)
    print(os.environ.get("LORA_AMD_WS_DEBUG","0"),M,K,N,round(med*1e6,2),flush=True)
