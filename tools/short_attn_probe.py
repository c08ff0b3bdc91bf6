import sys, os
sys.path.insert(0, os.getcwd())
import torch
from anyv2v_amd import ops
dev="cuda"
def t(fn, it=30, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it
for (B,HW,h) in ((3,4096,5),(3,1024,10),(3,256,20),(1,4096,5)):
    Fr=16; C=64*h
    qkv=torch.randn(B*Fr*HW,3*C,device=dev).half()
    o=torch.empty(B*Fr*HW,C,dtype=torch.float16,device=dev)
    st=(Fr*HW,1,HW)
    fn=lambda: ops.attention(qkv[:,:C],qkv[:,C:2*C],qkv[:,2*C:],o,batch=B*HW,heads=h,Sq=Fr,Sk=Fr,inner=HW,q_strides=st,kv_strides=st)
    ms=t(fn)
    gb=4*B*Fr*HW*C*2/1e9
    print(f"temporal attention B={B} HW={HW} h={h}: {ms*1e3:7.1f} us {gb/ms:6.2f} TB/s")
