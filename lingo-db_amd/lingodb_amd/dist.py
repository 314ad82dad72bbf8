"""Multi-GPU exchange of column buffers (one process per GPU, torch.distributed; backend "nccl" is
RCCL over xGMI on MI355X, "gloo" on CPU for the tests).

The data path has exactly two exchange shapes:
  * replicate a small relation on every rank (filtered dimension tables, partial aggregates):
    all-gather of variable-length column buffers;
  * re-partition a relation by the reference hash of its key (joins / high-cardinality group-bys):
    all-to-all of the buffers produced by ldb_gpu_partition.  On one MI355X node every peer pair
    has its own xGMI link, so the all-to-all is issued as one grouped batch of point-to-point
    sends/receives (ncclGroupStart … ncclSend/ncclRecv … ncclGroupEnd) rather than a ring.

Everything here operates on flat uint8 torch tensors (device-agnostic); the glue that wraps ldb
tables lives in tpch_dist.py.
"""
import torch


def exchange_counts(dist, send_counts, device):
    """send_counts[j] = rows this rank sends to rank j → recv_counts[i] = rows rank i sends here."""
    world = dist.get_world_size()
    mine = torch.tensor(list(send_counts), dtype=torch.int64, device=device)
    gathered = [torch.zeros(world, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(gathered, mine)
    rank = dist.get_rank()
    return [int(g[rank].item()) for g in gathered]


def allgather_columns(dist, cols, widths, n_rows):
    """Replicate: cols[k] = uint8 tensor holding n_rows * widths[k] bytes on every rank.
    Returns (list of concatenated uint8 tensors in rank order, per-rank row counts)."""
    world = dist.get_world_size()
    device = cols[0].device if cols else torch.device("cpu")
    cnt = torch.tensor([n_rows], dtype=torch.int64, device=device)
    counts_t = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(counts_t, cnt)
    counts = [int(c.item()) for c in counts_t]
    cap = max(max(counts), 1)
    # all columns travel in ONE collective: each rank's message is [col0 | col1 | …], every
    # column region padded to cap rows (small-message latency, not bandwidth, bounds these
    # exchanges on xGMI: a partial-aggregate table is a few KB in ~10 columns)
    region = [0]
    for w in widths:
        region.append(region[-1] + cap * w)
    packed = torch.zeros(max(region[-1], 1), dtype=torch.uint8, device=device)
    for k, (c, w) in enumerate(zip(cols, widths)):
        packed[region[k] : region[k] + n_rows * w] = c[: n_rows * w]
    parts = [torch.empty_like(packed) for _ in range(world)]
    dist.all_gather(parts, packed)
    out = []
    for k, w in enumerate(widths):
        out.append(torch.cat([p[region[k] : region[k] + counts[r] * w] for r, p in enumerate(parts)]))
    return out, counts


def alltoall_columns(dist, cols, widths, send_counts):
    """Re-partition: cols[k] holds the rows grouped by destination rank (send_counts[j] rows for
    rank j, in rank order — the layout ldb_gpu_partition produces).  Returns (received column
    tensors with the rows of rank 0, 1, … in that order, recv_counts)."""
    world, rank = dist.get_world_size(), dist.get_rank()
    device = cols[0].device if cols else torch.device("cpu")
    recv_counts = exchange_counts(dist, send_counts, device)
    send_off = [0]
    for c in send_counts:
        send_off.append(send_off[-1] + int(c))
    recv_off = [0]
    for c in recv_counts:
        recv_off.append(recv_off[-1] + int(c))
    # one message per peer pair carrying that peer's rows of EVERY column ([col0 rows | col1 rows | …]),
    # all pairs in one grouped batch: world-1 sends + world-1 receives per exchange
    total_w = sum(widths)
    ops, inbox = [], {}
    for peer in range(world):
        ns, nr = int(send_counts[peer]), recv_counts[peer]
        if peer == rank:
            continue
        if ns:
            msg = torch.cat([c[send_off[peer] * w : send_off[peer + 1] * w] for c, w in zip(cols, widths)]) if len(cols) > 1 else \
                cols[0][send_off[peer] * widths[0] : send_off[peer + 1] * widths[0]].contiguous()
            ops.append(dist.P2POp(dist.isend, msg, peer))
        if nr:
            inbox[peer] = torch.empty(nr * total_w, dtype=torch.uint8, device=device)
            ops.append(dist.P2POp(dist.irecv, inbox[peer], peer))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    outs = []
    col_off = 0  # byte offset of column k inside a peer's message, in units of that peer's row count
    for c, w in zip(cols, widths):
        pieces = []
        for peer in range(world):
            n = recv_counts[peer]
            if peer == rank:
                pieces.append(c[send_off[rank] * w : send_off[rank + 1] * w])
            elif n:
                pieces.append(inbox[peer][col_off * n : (col_off + w) * n])
        outs.append(torch.cat(pieces) if pieces else torch.empty(0, dtype=torch.uint8, device=device))
        col_off += w
    return outs, recv_counts


def shard_bounds(n_units, world):
    """contiguous unit ranges per rank (the row-range sharding of fact tables)"""
    return [(n_units * r // world, n_units * (r + 1) // world) for r in range(world)]
