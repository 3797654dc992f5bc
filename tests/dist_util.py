"""torch.distributed helpers of the measurement harness (the world_size-2 gloo tests on CPU; bench.py itself uses tools/rank_group.py).
They live outside the product package on purpose: network-slicing_amd/ never imports torch.  The process group is
only used to agree on wall-clock time (barrier, MAX over ranks) and to add up scalar reports."""
import numpy as np


def max_over_ranks(value, device=None):
    """MAX all-reduce of a python float (the timed region ends when the slowest rank ends)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(values, device=None):
    """SUM all-reduce of a small vector of counters."""
    import torch
    import torch.distributed as dist
    v = np.asarray(values, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return v
    t = torch.tensor(v, dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t.cpu().numpy()


def gather_exchange(device='cpu'):
    """exchange() callback for SharedVecKBRL over a torch.distributed process group: one all_gather of the proposal
    block.  The product's exchange is ncclAllGather inside libranslice.so (kb_shared_step); this host version exists
    so that the merge rule can be exercised by world_size-2 gloo tests on CPU."""
    import torch
    import torch.distributed as dist

    def ex(counts, props):
        W, me = dist.get_world_size(), dist.get_rank()
        blk = torch.from_numpy(np.concatenate([counts.astype(np.float64), props.ravel()])).to(device)
        out = [torch.empty_like(blk) for _ in range(W)]
        dist.all_gather(out, blk)
        arr = np.stack([o.cpu().numpy() for o in out])
        S = len(counts)
        return arr[:, :S].astype(np.int32), arr[:, S:].reshape((W,) + props.shape), me
    return ex
