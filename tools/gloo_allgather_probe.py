import os, sys, time, torch, torch.distributed as dist
dist.init_process_group('gloo', rank=int(os.environ['RANK']), world_size=int(os.environ['WORLD_SIZE']))
w=dist.get_world_size(); n=60416
send=torch.randn(n, dtype=torch.float64)
parts=[torch.zeros(n, dtype=torch.float64) for _ in range(w)]
for _ in range(3): dist.all_gather(parts, send)
t=time.time()
for _ in range(20): dist.all_gather(parts, send)
t1=(time.time()-t)/20
out=torch.zeros(n*w, dtype=torch.float64)
try:
    for _ in range(3): dist.all_gather_into_tensor(out, send)
    t=time.time()
    for _ in range(20): dist.all_gather_into_tensor(out, send)
    t2=(time.time()-t)/20
except Exception as e: t2=str(e)
buf=torch.zeros(n*w, dtype=torch.float64)
t=time.time()
for _ in range(20): dist.all_reduce(buf)
t3=(time.time()-t)/20
if dist.get_rank()==0: print('world',w,'all_gather list %.2f ms'%(t1*1e3), 'into_tensor', t2 if isinstance(t2,str) else '%.2f ms'%(t2*1e3), 'all_reduce %.2f ms'%(t3*1e3))
