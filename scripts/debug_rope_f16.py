import sys, numpy as np, torch
sys.path.insert(0, "vllm-backend-public_amd"); sys.path.insert(0, ".")
from mi355_decode import ops
from oracle import ops_ref as R
dtype="float16"; nh,kvh,d=4,2,64
rng=np.random.Generator(np.random.PCG64(1))
T,nblk,max_pos=8,4,64
table=R.rope_cos_sin_table(max_pos,d,5e5,dtype)
qkv=R.round_to(rng.standard_normal((T,(nh+2*kvh)*d)).astype(np.float32),dtype)
pos=rng.integers(0,max_pos,T).astype(np.int64)
slots=np.arange(T).astype(np.int64)
kc=torch.zeros(nblk,kvh,16,d,dtype=torch.float16,device="cuda"); vc=torch.zeros(nblk,kvh,d,16,dtype=torch.float16,device="cuda")
qd=torch.from_numpy(qkv).half().cuda()
ops.rope_kv_write(qd,torch.from_numpy(pos).cuda(),torch.from_numpy(table).half().cuda(),torch.from_numpy(slots).cuda(),kc,vc,nh,kvh,d)
got=qd.float().cpu().numpy()[:, :nh*d].reshape(T,nh,d)
x=qkv[:, :nh*d].reshape(T,nh,d); half=d//2
cs=table[pos]; cos=cs[:,None,:half]; sin=cs[:,None,half:]
x1,x2=x[...,:half],x[...,half:]
ref=R.rope_neox(x,pos,table,dtype)
single=np.concatenate([R.round_to(x1*cos-x2*sin,dtype),R.round_to(x2*cos+x1*sin,dtype)],-1)
f64=np.concatenate([R.round_to((x1.astype(np.float64)*cos-x2.astype(np.float64)*sin).astype(np.float32),dtype),R.round_to((x2.astype(np.float64)*cos+x1.astype(np.float64)*sin).astype(np.float32),dtype)],-1)
print("mismatch vs oracle(3 roundings):",(got!=ref).mean())
print("mismatch vs single rounding fp32:",(got!=single).mean())
print("mismatch vs single rounding f64:",(got!=f64).mean())
# partial: round only first product
p1=np.concatenate([R.round_to(R.round_to(x1*cos,dtype)-x2*sin,dtype),R.round_to(R.round_to(x2*cos,dtype)+x1*sin,dtype)],-1)
p2=np.concatenate([R.round_to(x1*cos-R.round_to(x2*sin,dtype),dtype),R.round_to(x2*cos+R.round_to(x1*sin,dtype),dtype)],-1)
print("mismatch vs round-first-only:",(got!=p1).mean()," round-second-only:",(got!=p2).mean())
