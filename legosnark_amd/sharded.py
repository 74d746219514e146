"""Multi-GPU MSM: contiguous index ranges per rank, one exchange step.

libff's multi_exp splits [0, n) into `chunks` contiguous ranges (the last takes the
remainder), runs multi_exp_inner per chunk and sums the partials
(/root/reference/src/utils/globl.h:67-77 forwards `chunks`).  Here a
chunk is a GPU: every rank runs the single-GPU Pippenger on its slice, the 96-byte (G1) /
192-byte (G2) Jacobian partials are exchanged with ONE all-gather (RCCL over xGMI when
the backend is "nccl"; RCCL has no elliptic-curve reduce op, so reduce = gather + local
fold), and the partials are summed in rank order.  Group addition is associative and
commutative, so the result equals the 1-GPU result after affine normalisation.

The compute callbacks default to the HIP library; tests inject a stand-in so that the
partitioning / collective / fold logic runs under gloo on CPU.
"""
import numpy as np


def shard_range(n, world, rank):
    """libff multi_exp chunking: one = n // world; the last rank takes the remainder."""
    if world <= 1 or n < world:
        return (0, n) if rank == 0 else (n, n)
    one = n // world
    lo = rank * one
    hi = n if rank == world - 1 else lo + one
    return lo, hi


class ShardedMSM:
    """One instance per rank.  `local_msm(d_scalars, d_out)` writes this rank's partial
    (Jacobian limbs, int64[w]) into d_out; `fold(gathered, world, d_total)` sums the
    gathered partials.

    With `stream` (the library's HIP stream) and `side` (a second stream) the exchange step of
    call i -- wait for the MSM tail, all-gather, fold -- runs on `side`, so the library stream
    is free to start the front of call i+1: the pipelining a single GPU gets from the
    internal tail streams survives the collective.  Buffers rotate over DEPTH calls."""

    DEPTH = 4

    def __init__(self, group, world, rank, local_msm, fold, device, dist=None, stream=None, side=None, join_side=None,
                 join_main=None, sync=None):
        import torch
        self.torch = torch
        self.group = group
        self.w = 12 if group == "g1" else 24
        self.world, self.rank = world, rank
        self.local_msm, self.fold = local_msm, fold
        self.dist = dist
        self.stream, self.side, self.join_side = stream, side, join_side
        # join_main: orders the library stream after the MSM tails issued so far (lsa_stream_join);
        # sync: the same plus a host-side wait (lsa_synchronize)
        self.join_main, self.sync = join_main, sync
        # collective form, chosen once: RCCL has all_gather_into_tensor; gloo only the list form
        backend = dist.get_backend() if (dist is not None and hasattr(dist, "get_backend")) else "nccl"
        self.tensor_form = backend == "nccl"
        depth = self.DEPTH if side is not None else 1
        self.partial = [torch.zeros(self.w, dtype=torch.int64, device=device) for _ in range(depth)]
        self.gathered = [torch.zeros((world, self.w), dtype=torch.int64, device=device) for _ in range(depth)]
        self.total = [torch.zeros(self.w, dtype=torch.int64, device=device) for _ in range(depth)]
        self.done = [None] * depth
        self.calls = 0

    def run(self, d_scalars):
        """Asynchronous; returns the device tensor holding the sum (identical on every rank);
        read it through result_host()."""
        j = self.calls % len(self.partial)
        self.calls += 1
        if self.world > 1 and self.side is not None and self.done[j] is not None:
            self.stream.wait_event(self.done[j])         # buffer set j was last used DEPTH calls ago
        self.local_msm(d_scalars, self.partial[j])
        if self.world == 1:
            return self.partial[j]
        if self.side is not None:
            self.side.wait_stream(self.stream)            # front of this call (and everything before it)
            self.join_side(self.side)                     # ... and its tail, without stalling the library stream
            with self.torch.cuda.stream(self.side):
                self._all_gather(j)
                self.fold(self.gathered[j], self.world, self.total[j], self.side)
                self.done[j] = self.side.record_event()
        elif self.stream is not None:
            cur = self.torch.cuda.current_stream()
            if self.join_main is not None:
                self.join_main()                          # the tail that publishes the partial
            cur.wait_stream(self.stream)                  # partial is ready
            self._all_gather(j)
            self.stream.wait_stream(cur)                  # gathered is ready
            self.fold(self.gathered[j], self.world, self.total[j], None)
        else:
            self._all_gather(j)
            self.fold(self.gathered[j], self.world, self.total[j], None)
        return self.total[j]

    def _all_gather(self, j):
        if self.tensor_form:
            self.dist.all_gather_into_tensor(self.gathered[j].view(-1), self.partial[j])
        else:
            # backends without the tensor form (gloo): list form, same data movement
            parts = [self.gathered[j][i] for i in range(self.world)]
            self.dist.all_gather(parts, self.partial[j])

    def result_host(self, d_result):
        if self.side is not None:
            self.side.synchronize()
        if self.sync is not None:
            self.sync()                                   # MSM tails (internal streams) + library stream
        elif self.stream is not None:
            self.stream.synchronize()
        return d_result.cpu().numpy().view(np.uint64).copy()


def make_gpu_sharded(lsa, group, bases_handle, world, rank, dist=None):
    """Wire ShardedMSM to the HIP library: local MSM over the rank's device-resident bases,
    the all-gather (RCCL when the backend is "nccl") and the fold with lsa_*_sum on a side
    stream that waits for the MSM's tail."""
    import torch
    dev = torch.device("cuda", torch.cuda.current_device())
    ext = torch.cuda.ExternalStream(lsa.stream_handle(), device=dev)
    side = torch.cuda.Stream(device=dev) if world > 1 else None

    def local_msm(d_scalars, d_out):
        bases_handle.msm_async(d_scalars, d_out)

    def fold(gathered, n, d_total, stream):
        if stream is None:
            lsa.sum_async(group, gathered, n, d_total)
        else:
            lsa.sum_on(group, gathered, n, d_total, stream.cuda_stream)

    def join_side(stream):
        lsa.stream_join_to(stream.cuda_stream)

    return ShardedMSM(group, world, rank, local_msm, fold, dev, dist=dist, stream=ext, side=side, join_side=join_side,
                      join_main=lsa.stream_join, sync=lsa.synchronize)


class ShardedPairingProduct:
    """final_exponentiation(prod_i miller_loop(P_i, Q_i)) with the batch split over ranks
    ("Pairings"): rank r runs the Miller loops of its contiguous slice
    and multiplies them locally, the 384-byte Fq12 partial products are all-gathered, every
    rank multiplies the `world` partials in rank order and runs the one final exponentiation.
    Fq12 multiplication is commutative, so the GT value equals the 1-GPU value bit for bit.

    `local_product(g1, g2)` -> (48,) uint64 partial; `product(f)` -> (48,) uint64 product of
    the rows of f; `final_exp(f)` -> (48,) uint64.  The defaults are the HIP library's
    lsa_miller_loop_product / lsa_fq12_product / lsa_final_exponentiation."""

    def __init__(self, world, rank, local_product, product, final_exp, dist=None, device="cpu"):
        import torch
        self.torch = torch
        self.world, self.rank = world, rank
        self.local_product, self.product, self.final_exp = local_product, product, final_exp
        self.dist = dist
        self.device = device

    def run(self, g1, g2):
        g1 = np.ascontiguousarray(g1, dtype=np.uint64).reshape(-1, 12)
        g2 = np.ascontiguousarray(g2, dtype=np.uint64).reshape(-1, 24)
        if len(g1) != len(g2):
            raise ValueError("need as many G1 as G2 points")
        lo, hi = shard_range(len(g1), self.world, self.rank)
        partial = self.local_product(g1[lo:hi], g2[lo:hi])     # Fq12 one for an empty slice
        if self.world > 1:
            t = self.torch.from_numpy(partial.view(np.int64).copy()).to(self.device)
            parts = [self.torch.empty_like(t) for _ in range(self.world)]
            self.dist.all_gather(parts, t)
            gathered = np.stack([p.cpu().numpy().view(np.uint64) for p in parts])
            partial = self.product(gathered)
        return self.final_exp(partial)


def make_gpu_sharded_pairing(lsa, world, rank, dist=None):
    import torch
    dev = torch.device("cuda", torch.cuda.current_device()) if (dist is not None and dist.get_backend() == "nccl") else "cpu"
    return ShardedPairingProduct(world, rank, lsa.miller_loop_product, lsa.fq12_product,
                                 lambda f: lsa.final_exponentiation(f)[0], dist=dist, device=dev)
