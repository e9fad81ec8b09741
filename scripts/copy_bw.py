import torch, time
dev=torch.device("cuda",0)
for gib in (1,4,16):
    n=gib*(1<<30)//8
    a=torch.empty(n,dtype=torch.int64,device=dev); b=torch.empty_like(a)
    a.fill_(3); torch.cuda.synchronize()
    for _ in range(2): b.copy_(a)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): b.copy_(a)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/5
    print(f"copy {gib} GiB: {ms:.3f} ms  read+write {2*gib*1.0737/ms:.2f} TB/s")
    e0.record()
    for _ in range(5): b.fill_(7)
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/5
    print(f"fill {gib} GiB: {ms:.3f} ms  write {gib*1.0737/ms:.2f} TB/s")
    e0.record()
    for _ in range(5): s=a.sum()
    e1.record(); torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/5
    print(f"sum  {gib} GiB: {ms:.3f} ms  read {gib*1.0737/ms:.2f} TB/s")
    del a,b
