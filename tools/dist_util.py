"""torch.distributed helpers of the measurement harness (bench.py, tools/run_shared_kbrl.py and the gloo tests).
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
