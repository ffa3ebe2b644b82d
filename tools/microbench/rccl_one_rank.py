"""What a torch.distributed collective over RCCL costs on ONE rank (no wire): python tools/microbench/rccl_one_rank.py.
Under rocprofv3 each call shows up as ~1.7 buffer copies + ~2.5 buffer fills -- RCCL's own one-rank path, not this
package's; DESIGN.md section 6 quotes the per-call times next to the data-parallel step's 22-24 us."""
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29577', RANK='0', WORLD_SIZE='1')
dist.init_process_group('nccl', rank=0, world_size=1)
g = torch.zeros(182402, device='cuda'); out = torch.zeros(182402, device='cuda')
for _ in range(5): dist.all_reduce(g)
torch.cuda.synchronize()
import time
t0=time.perf_counter()
for _ in range(100): dist.all_reduce(g)
torch.cuda.synchronize(); t1=time.perf_counter()
for _ in range(100): dist.all_gather_into_tensor(out, g)
torch.cuda.synchronize(); t2=time.perf_counter()
print('all_reduce %.1f us  all_gather %.1f us per call (1 rank, 0.73 MB)' % ((t1-t0)*1e4, (t2-t1)*1e4))
dist.destroy_process_group()
