"""Multi-GPU layout of the hot path: (pair, window) units are independent (processor.go:816-850
is a plain double loop), so windows are dealt round-robin to ranks -- each rank transforms every
station's window once and reuses the spectrum for all pairs -- and the only exchange is one
all-gather of the fixed-size per-pair peak records (RCCL over xGMI on GPUs, gloo in CPU tests).
"""
import numpy as np

from .capi import PEAK_DTYPE

PEAK_BYTES = PEAK_DTYPE.itemsize  # 16


def owned_windows(rank, world, n_windows):
    """Window ids processed by `rank` (window-major sharding: wid % world == rank)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    return list(range(rank, n_windows, world))


def owner_of(wid, world):
    return wid % world


def unit_owner(wid, pair, world, n_windows, n_pairs):
    """Rank that computes pair `pair` of window `wid` in tdoa_process(rank, world): window-major, or -- with fewer
    windows than ranks -- (window, pair) units dealt round-robin (SURVEY section 8e fallback; a rank then transforms
    only the stations its pairs need)."""
    if n_windows < world:
        return (wid * n_pairs + pair) % world
    return wid % world


def window_grid(n_samples, window_len):
    """(block, window length, windows per block) of a capture of n_samples: own thirds per file (processor.go:214),
    windows of window_len inside each block (processor.go:772-780 generalised) -- the library's window_geometry"""
    block = n_samples // 3
    wlen = min(window_len, block)
    return block, wlen, max(1, block // wlen) if wlen else 0


def owned_sample_runs(rank, world, n_samples, window_len, n_min=None):
    """[(first_sample, n_samples)] runs of a capture that tdoa_process(rank, world) reads under window-major sharding
    (adjacent owned windows merged).  With fewer windows than ranks the pair-major fallback may touch any window:
    then the whole capture is one run.  n_min: length of the shortest capture of the job when they differ (the
    window grid comes from it, the block offsets from the capture's own thirds)."""
    block = n_samples // 3
    _, wlen, wpb = window_grid(n_samples if n_min is None else n_min, window_len)
    n_windows = 3 * wpb
    if n_windows < world:
        return [(0, n_samples)]
    runs = []
    for wid in owned_windows(rank, world, n_windows):
        first = (wid // wpb) * block + (wid % wpb) * wlen
        if runs and runs[-1][0] + runs[-1][1] == first:
            runs[-1] = (runs[-1][0], runs[-1][1] + wlen)
        else:
            runs.append((first, wlen))
    return runs


def peaks_as_bytes(peaks):
    """structured peak array -> flat uint8 view (what travels through the collective)."""
    a = np.ascontiguousarray(peaks, dtype=PEAK_DTYPE)
    return a.view(np.uint8).reshape(-1)


def bytes_as_peaks(buf, n_windows, n_pairs):
    return np.frombuffer(np.ascontiguousarray(buf, dtype=np.uint8).tobytes(), dtype=PEAK_DTYPE).reshape(
        n_windows, n_pairs).copy()


def all_gather_peaks(local_bytes, dist, group=None):
    """One all-gather of this rank's peak buffer (torch uint8 tensor, CPU or GPU).
    Returns a [world, nbytes] tensor on every rank."""
    import torch
    world = dist.get_world_size(group)
    out = torch.empty((world, local_bytes.numel()), dtype=torch.uint8, device=local_bytes.device)
    if local_bytes.is_cuda:
        dist.all_gather_into_tensor(out.view(-1), local_bytes.contiguous(), group=group)
    else:  # gloo: list form
        parts = [out[r] for r in range(world)]
        dist.all_gather(parts, local_bytes.contiguous(), group=group)
    return out


def merge_sharded(gathered, n_windows, n_pairs):
    """gathered[r] = rank r's [n_windows][n_pairs] peaks with other ranks' units zero-filled
    (tdoa_process(rank, world)); returns the complete array, each unit taken from its owner."""
    world = gathered.shape[0]
    out = np.zeros((n_windows, n_pairs), dtype=PEAK_DTYPE)
    parts = [bytes_as_peaks(np.asarray(gathered[r].cpu() if hasattr(gathered[r], "cpu") else gathered[r]),
                            n_windows, n_pairs) for r in range(world)]
    for wid in range(n_windows):
        for p in range(n_pairs):
            out[wid, p] = parts[unit_owner(wid, p, world, n_windows, n_pairs)][wid, p]
    return out
