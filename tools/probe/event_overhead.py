import torch
torch.cuda.init()
x=torch.zeros(1024,device='cuda')
s=torch.cuda.current_stream()
def pairs(n, work):
    out=[]
    for _ in range(n):
        e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
        e0.record()
        if work: x.add_(1)
        e1.record()
        out.append((e0,e1))
    torch.cuda.synchronize()
    v=sorted(a.elapsed_time(b)*1e3 for a,b in out)
    return v[len(v)//2], v[0], v[-1]
for w in (False, True):
    print('work' if w else 'empty', pairs(64,w))
# busy stream: queue long work first so events are processed back-to-back on the GPU
big=torch.zeros(64*1024*1024,device='cuda')
for w in (False, True):
    for _ in range(20): big.add_(1)
    print('behind queue', 'work' if w else 'empty', pairs(64,w))
