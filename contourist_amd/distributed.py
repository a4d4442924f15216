"""Slab-partitioned extraction over several GPUs of one node (one process per GPU).

The volume is cut into contiguous slabs along array axis 0 (the reference's x, BASELINE.json's
"z-slab").  Voxels are independent given their corners, so the only exchange on the data path is
ONE sample plane per slab boundary: rank r receives the first plane of rank r+1 as its halo
(`torch.distributed` send/recv: RCCL over xGMI with backend "nccl", gloo on CPU tensors).
Edge ids are global by formula -- (global linear index of the lower lattice point << 3) | direction
-- so slab meshes concatenate without any renumbering exchange; vertices that a slab sees only
through its halo plane are owned (and emitted) by the upper neighbour.

There is no counterpart in the reference (it is single-process Python); the slab rule follows
SURVEY.md section 8e.
"""
import numpy as np


def _ffi_flag_march():
    from . import _ffi
    return _ffi.CX_MESH_OF_THE_MARCH


def slab_bounds(n0, world, rank):
    "planes [i0, i1) of axis 0 owned by `rank`"
    return (rank * n0) // world, ((rank + 1) * n0) // world


def exchange_halo(local, n_own, rank, world, dist=None):
    """local: tensor of n_own (+1 if rank < world-1) planes.  Sends the first owned plane to
    rank-1 and receives the halo plane local[n_own] from rank+1."""
    if world == 1:
        return
    if dist is None:
        import torch.distributed as dist
    staged = local.is_cuda and dist.get_backend() == "gloo"      # gloo moves host memory: stage the plane
    send = local[0].contiguous()
    recv = local[n_own] if rank + 1 < world else None
    if staged:
        send = send.cpu()
        recv = recv.cpu() if recv is not None else None
    ops = []
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, send, rank - 1))
    if rank + 1 < world:
        ops.append(dist.P2POp(dist.irecv, recv, rank + 1))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if staged and recv is not None:
        local[n_own].copy_(recv)


class HaloExchange(object):
    """a halo exchange in flight: start() posts the send / receive, finish() makes the current stream (and, for
    host-staged gloo, the host) wait for it.  Lets the exchange for the NEXT volume run while the current one
    is being extracted."""

    def __init__(self, local, n_own, rank, world, dist=None):
        if dist is None:
            import torch.distributed as dist
        self.local, self.n_own, self.works, self.staged_recv = local, n_own, [], None
        if world == 1:
            return
        staged = local.is_cuda and dist.get_backend() == "gloo"      # gloo moves host memory: stage the plane
        send = local[0].contiguous()
        recv = local[n_own] if rank + 1 < world else None
        if staged:
            send = send.cpu()
            recv = recv.cpu() if recv is not None else None
            self.staged_recv = recv
        ops = []
        if rank > 0:
            ops.append(dist.P2POp(dist.isend, send, rank - 1))
        if rank + 1 < world:
            ops.append(dist.P2POp(dist.irecv, recv, rank + 1))
        self._keep = (send, recv)
        if ops:
            self.works = dist.batch_isend_irecv(ops)

    def finish(self):
        for w in self.works:
            w.wait()
        self.works = []
        if self.staged_recv is not None:
            self.local[self.n_own].copy_(self.staged_recv)
            self.staged_recv = None


def own_communicators(ctxs, rank, world, dist=None):
    """Give this rank's contexts ONE RCCL communicator (cx_rccl_comm_init on the first, cx_rccl_comm_share on the others), so that
    the halo exchange of a step is part of ONE C call (Context.slab_step) instead of a Python batch_isend_irecv.  The 128-byte id
    travels by torch.distributed (whatever backend it runs).  Collective; returns True on EVERY rank only if every rank succeeded
    (otherwise the callers stay on exchange_halo / HaloExchange) -- and nobody enters the blocking collective ncclCommInitRank
    unless every rank has said, after checking everything that can fail locally, that it will enter it too."""
    import torch
    if dist is None:
        import torch.distributed as dist
    if world == 1:
        return True
    on_device = dist.get_backend() == "nccl"
    dev = torch.device("cuda", torch.cuda.current_device()) if on_device else torch.device("cpu")

    def all_agree(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    # 1. everything local: RCCL resolves in this process, the id can be made (rank 0), every context is alive
    ok, uid = True, np.zeros(128, dtype=np.uint8)
    try:
        ok = len(ctxs) > 0 and all(getattr(c, "handle", None) for c in ctxs) and ctxs[0].rccl_available()
        if ok and rank == 0:
            uid = ctxs[0].rccl_unique_id()
    except Exception:       # noqa: BLE001 -- reported through the agreed flag
        ok = False
    if not all_agree(ok):
        return False
    t = torch.from_numpy(np.ascontiguousarray(uid).copy()).to(dev)
    dist.broadcast(t, src=0)
    # 2. the collective init, on ONE context; the others of this rank share its communicator
    try:
        ctxs[0].rccl_comm_init(t.cpu().numpy(), rank, world)
        for c in ctxs[1:]:
            c.rccl_comm_share(ctxs[0])
    except Exception:       # noqa: BLE001
        ok = False
    if not all_agree(ok):
        for c in ctxs:          # nobody keeps half a setup: every rank falls back together
            try:
                c.rccl_comm_destroy()
            except Exception:   # noqa: BLE001
                pass
        return False
    return True


def hip_extract(device=0, diagonal_flags=1, float64_points=False, context=None):
    """default local extractor: the HIP Level-0 march. returns f(local_array_or_tensor, value, origin) -> (xyz, keys, tris).
    float64_points: xyz are the float64 coordinates the reference interpolates (what the Level-1 post-pass works on)
    instead of the march's fp32 ones.  run.context is the context used."""
    from . import _ffi
    ctx = context or _ffi.Context(device)

    def run(local, value, origin=(0, 0, 0)):
        ctx.set_origin(*origin)
        if type(local).__module__.split(".")[0] == "torch":
            ctx.adopt_device_grid(local.data_ptr(), tuple(local.shape), keepalive=local)
        else:
            ctx.upload_grid(local)
        counts = ctx.extract3d(value, diagonal_flags)
        xyz, keys, tris = ctx.download_level0(counts)
        if float64_points:
            xyz = ctx.level0_points_f64(counts)
        return xyz, keys, tris
    run.context = ctx
    run.global_points = bool(float64_points)   # level0_points_f64 already adds the origin (exactly, before interpolating)
    return run


def local_to_global(xyz, keys, tris, local_shape, i0, n_own, has_halo, xyz_is_global=False):
    """Level-0 mesh of one slab -> global ids.
    returns (gkeys (V',) int64 of the vertices this rank OWNS, gxyz (V',3), tri_gkeys (T,3) int64)"""
    n1, n2 = int(local_shape[1]), int(local_shape[2])
    keys = np.asarray(keys).astype(np.int64)
    offset = (np.int64(i0) * n1 * n2) << 3
    gkeys = keys + offset
    tri_gkeys = gkeys[np.asarray(tris, dtype=np.int64)] if len(tris) else np.zeros((0, 3), dtype=np.int64)
    gxyz = np.asarray(xyz, dtype=np.float64).copy()
    if not xyz_is_global:
        gxyz[:, 0] += i0
    if has_halo:
        owner_plane = (keys >> 3) // (n1 * n2)           # local plane of the owning lattice point
        own = owner_plane < n_own
        gkeys, gxyz = gkeys[own], gxyz[own]
    return gkeys, gxyz, tri_gkeys


def assemble(parts):
    """parts: list of (gkeys, gxyz, tri_gkeys) from all ranks -> (keys sorted, xyz, triangles as indices)"""
    keys = np.concatenate([p[0] for p in parts]) if parts else np.zeros(0, np.int64)
    xyz = np.concatenate([p[1] for p in parts]) if parts else np.zeros((0, 3))
    tk = np.concatenate([p[2] for p in parts]) if parts else np.zeros((0, 3), np.int64)
    order = np.argsort(keys, kind="stable")
    keys, xyz = keys[order], xyz[order]
    if len(keys) > 1 and np.any(keys[1:] == keys[:-1]):
        raise RuntimeError("a vertex was emitted by two slabs")
    tris = np.searchsorted(keys, tk)
    if len(tk) and not np.array_equal(keys[tris], tk):
        raise RuntimeError("a triangle references a vertex no slab emitted")
    return keys, xyz, tris


def extract_slabs(own_planes, value, rank, world, extract_fn, global_shape, dist=None, gather=True):
    """own_planes: this rank's planes [i0, i1) (numpy array or torch tensor, on CPU or GPU).
    Runs halo exchange + local march + global id conversion; with gather=True rank 0 returns the
    assembled (keys, xyz, triangles), other ranks None."""
    import torch
    if dist is None:
        import torch.distributed as dist
    n0 = int(global_shape[0])
    i0, i1 = slab_bounds(n0, world, rank)
    n_own = i1 - i0
    has_halo = rank + 1 < world
    t = own_planes if isinstance(own_planes, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(own_planes, dtype=np.float32))
    assert t.shape[0] == n_own, (t.shape, n_own)
    local = torch.empty((n_own + (1 if has_halo else 0),) + tuple(t.shape[1:]), dtype=torch.float32, device=t.device)
    local[:n_own] = t
    exchange_halo(local, n_own, rank, world, dist)
    arg = local if local.is_cuda else local.numpy()
    xyz, keys, tris = extract_fn(arg, value, (i0, 0, 0))
    part = local_to_global(xyz, keys, tris, tuple(local.shape), i0, n_own, has_halo, getattr(extract_fn, "global_points", False))
    if not gather:
        return part
    if world == 1:
        return assemble([part])
    gathered = gather_parts(part, rank, world, dist, local.device if local.is_cuda else None)
    if rank == 0:
        return assemble(gathered)
    return None


def hip_extract_levels(device=0, diagonal_flags=1, context=None):
    """local extractor for several isovalues of one slab in ONE call (cx_extract3d_levels: the samples are streamed once for
    all levels): f(local_array_or_tensor, values, origin) -> list of (xyz, keys, tris), one per value.  run.context is the
    context used."""
    from . import _ffi
    ctx = context or _ffi.Context(device)

    def run(local, values, origin=(0, 0, 0)):
        ctx.set_origin(*origin)
        if type(local).__module__.split(".")[0] == "torch":
            ctx.adopt_device_grid(local.data_ptr(), tuple(local.shape), keepalive=local)
        else:
            ctx.upload_grid(local)
        counts = ctx.extract3d_levels(values, diagonal_flags)
        out = []
        for l, c in enumerate(counts):
            ctx.select_level(l)
            out.append(ctx.download_level0(c))
        return out
    run.context = ctx
    run.global_points = False
    return run


def extract_slabs_levels(own_planes, values, rank, world, extract_levels_fn, global_shape, dist=None, gather=True):
    """BASELINE config 5 across ranks: every rank marches ITS slab (own planes + one halo plane) for ALL isovalues in one
    call -- the slab is streamed once however many levels there are (multiple_2d_contour.py:17-30, 48-59 classifies a value
    against all sorted levels at once; here in 3-D) -- and the levels' meshes are assembled level by level exactly as
    extract_slabs does for one.  extract_levels_fn(local, values, origin) -> [(xyz, keys, tris)] per value
    (hip_extract_levels; tests pass the oracle).  gather=True: rank 0 returns [(keys, xyz, triangles)] per value, others None;
    gather=False: every rank returns its parts [(gkeys, gxyz, tri_gkeys)] per value."""
    import torch
    if dist is None:
        import torch.distributed as dist
    n0 = int(global_shape[0])
    i0, i1 = slab_bounds(n0, world, rank)
    n_own = i1 - i0
    has_halo = rank + 1 < world
    t = own_planes if isinstance(own_planes, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(own_planes, dtype=np.float32))
    assert t.shape[0] == n_own, (t.shape, n_own)
    local = torch.empty((n_own + (1 if has_halo else 0),) + tuple(t.shape[1:]), dtype=torch.float32, device=t.device)
    local[:n_own] = t
    exchange_halo(local, n_own, rank, world, dist)                     # ONE halo plane serves every level
    arg = local if local.is_cuda else local.numpy()
    meshes = extract_levels_fn(arg, [float(v) for v in values], (i0, 0, 0))
    glob = getattr(extract_levels_fn, "global_points", False)
    parts = [local_to_global(xyz, keys, tris, tuple(local.shape), i0, n_own, has_halo, glob) for xyz, keys, tris in meshes]
    if not gather:
        return parts
    if world == 1:
        return [assemble([p]) for p in parts]
    out = []
    for p in parts:
        g = gather_parts(p, rank, world, dist, local.device if local.is_cuda else None)
        out.append(assemble(g) if rank == 0 else None)
    return out if rank == 0 else None


def gather_parts(part, rank, world, dist, device=None):
    """(gkeys (V,) int64, gxyz (V,3) float64, tri_gkeys (T,3) int64) of every rank -> list of them on rank 0 (None elsewhere).
    Tensors, not pickles: the sizes go round with one all_gather, then every rank sends ONE flat 8-byte-word buffer
    (keys | coordinates as their bit patterns | triangle keys) to rank 0 -- over RCCL (device buffers) with backend "nccl",
    host buffers with gloo.  A 512^3 surface is ~0.5 GB per volume: this is what `gather_object` used to pickle."""
    import torch
    gkeys, gxyz, tk = part
    gkeys = np.ascontiguousarray(gkeys, dtype=np.int64).reshape(-1)
    gxyz = np.ascontiguousarray(gxyz, dtype=np.float64).reshape(-1, 3)
    tk = np.ascontiguousarray(tk, dtype=np.int64).reshape(-1, 3)
    on_device = dist.get_backend() == "nccl"
    dev = device if (on_device and device is not None) else (torch.device("cuda", torch.cuda.current_device()) if on_device else torch.device("cpu"))
    sizes = torch.tensor([len(gkeys), len(tk)], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes)
    all_sizes = [tuple(int(x) for x in t.cpu()) for t in all_sizes]
    flat = np.concatenate([gkeys, gxyz.reshape(-1).view(np.int64), tk.reshape(-1)])
    if rank != 0:
        if len(flat):
            dist.send(torch.from_numpy(flat).to(dev), dst=0)
        return None
    out = [part]
    for r in range(1, world):
        nv, nt = all_sizes[r]
        n = nv * 4 + nt * 3
        buf = torch.empty(n, dtype=torch.int64, device=dev)
        if n:
            dist.recv(buf, src=r)
        h = buf.cpu().numpy()
        out.append((h[:nv].copy(), h[nv:nv * 4].view(np.float64).reshape(-1, 3).copy(), h[nv * 4:].reshape(-1, 3).copy()))
    return out


def level1_slabs(own_planes, value, rank, world, global_shape, device=0, clean=True, smooth=None, dist=None):
    """Level 1 (weld, tiny collapse, clean, orient: tetrahedral.py:190-215, 353-375, surface_geometry.py:14-140) of a
    volume that is spread over the ranks in slabs along axis 0: every rank marches its slab on its GPU, the Level-0
    meshes are gathered on rank 0 (vertices with the float64 coordinates the reference interpolates, global edge ids)
    and post-processed there as ONE mesh -- components, weld buckets and the max-x orientation rule see the whole
    surface.  Rank 0 returns (grid_points (V,3) float64, triangles (T,3) int32, counts), the others None."""
    fn = hip_extract(device, float64_points=True)
    mesh = extract_slabs(own_planes, value, rank, world, fn, global_shape, dist=dist, gather=True)
    if rank != 0:
        return None
    keys, xyz, tris = mesh
    ctx = fn.context
    corner = [int(n) - 1 for n in global_shape]
    post = ctx.postprocess3d_mesh(xyz, tris, corner, (0 if clean else 1) | _ffi_flag_march(), smooth or 0.0)
    pts, t1 = ctx.download_level1(post)
    return pts, t1, post


# ---- Level 1 sharded over the slabs -----------------------------------------------------------------------
# The reference post-processes one mesh in one process (tetrahedral.py:190-215 weld, :353-375 tiny collapse,
# surface_geometry.py:14-50 clean-up, :52-140 orientation).  Here every rank post-processes ITS slab on its GPU: the slab is
# marched together with SHARD_LAYERS layers of cells of each neighbour, which is all that weld buckets (narrower than a cell,
# never across an integer plane), tiny and degenerate triangles (chains around one lattice point) can see of the neighbours;
# only the orientation is global -- components and the max-x start triangle (surface_geometry.py:71-103) -- and for that the
# ranks exchange what lies at the slab boundaries: the labels of the triangles next to a boundary and one start-triangle
# candidate per component that reaches one.  Rank 0 unites the labels and picks the candidates (work ~ boundary size).
SHARD_LAYERS = 2


def exchange_planes(own, rank, world, below, above, dist=None):
    """own: this rank's planes (tensor, n_own x n1 x n2).  -> (local, n_below): `local` = up to `below` last planes of rank-1,
    the own planes, up to `above` first planes of rank+1 (every slab must hold that many planes)."""
    import torch
    if dist is None:
        import torch.distributed as dist
    n_own = int(own.shape[0])
    nb = below if rank > 0 else 0
    na = above if rank + 1 < world else 0
    local = torch.empty((nb + n_own + na,) + tuple(own.shape[1:]), dtype=own.dtype, device=own.device)
    local[nb:nb + n_own] = own
    if world == 1:
        return local, 0
    assert n_own >= max(below, above), "slabs thinner than the exchanged layers"
    staged = own.is_cuda and dist.get_backend() == "gloo"
    up = own[n_own - below:].contiguous()        # what rank+1 sees below itself
    down = own[:above].contiguous()              # what rank-1 sees above itself
    rb = torch.empty((nb,) + tuple(own.shape[1:]), dtype=own.dtype, device="cpu" if staged else own.device)
    ra = torch.empty((na,) + tuple(own.shape[1:]), dtype=own.dtype, device="cpu" if staged else own.device)
    if staged:
        up, down = up.cpu(), down.cpu()
    ops = []
    if rank > 0:
        ops.append(dist.P2POp(dist.isend, down, rank - 1))
        ops.append(dist.P2POp(dist.irecv, rb, rank - 1))
    if rank + 1 < world:
        ops.append(dist.P2POp(dist.isend, up, rank + 1))
        ops.append(dist.P2POp(dist.irecv, ra, rank + 1))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    if nb:
        local[:nb].copy_(rb)
    if na:
        local[nb + n_own:].copy_(ra)
    return local, nb


def pair_labels(own_hash, own_label, copy_hash, copy_label):
    """a rank's own triangles next to its lower neighbour (list 1 of Context.shard_boundary) against that neighbour's copies of
    them (its list 4): torch tensors on one device.  -> (pairs (P,2) int64 numpy, distinct [own label, neighbour's label],
    unmatched: triangles only one side knows).  The lists hold the same triangles; sorted by hash they line up."""
    import torch
    no = int(own_hash.numel())
    nc = int(copy_hash.numel())
    if no == 0 or nc == 0:
        return np.zeros((0, 2), dtype=np.int64), no + nc
    ho, io = torch.sort(own_hash)
    lo = own_label[io].to(torch.int64)
    at = torch.searchsorted(ho, copy_hash).clamp_(max=no - 1)
    ok = ho[at] == copy_hash
    seen = torch.zeros(no, dtype=torch.bool, device=ho.device)
    seen[at[ok]] = True
    dup = int((ho[1:] == ho[:-1]).sum().item()) if no > 1 else 0          # two triangles with one hash: their labels cannot be told apart
    unmatched = int((~ok).sum().item()) + int((~seen).sum().item()) + dup
    packed = torch.unique((lo[at[ok]] << 32) | copy_label[ok].to(torch.int64))
    packed = packed.cpu().numpy()
    return np.stack([packed >> 32, packed & 0xFFFFFFFF], axis=1).astype(np.int64), unmatched


def merge_shard_components(lists):
    """rank 0's part of the sharded orientation.  lists[r] = dict(pairs (P,2) [label on rank r, label on rank r-1] of components
    that are one (pair_labels), cand_label, cand_x, cand_vertex_key (edge ids of the WHOLE volume, int64), cand_nx,
    cand_negative, cand_has) of rank r: sizes follow the number of components at the slab boundaries.
    -> ([(labels, flips) per rank], stats).  Per component the start triangle is the candidate with the largest
    (x, vertex edge id, |normal_x|) -- surface_geometry.py:79-94 with the ties broken by edge id -- and the component is flipped
    iff that triangle's normal_x is negative (:99-103)."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components
    world = len(lists)
    uniq, base = [], [0]
    for r, L in enumerate(lists):
        mine = [np.asarray(L["cand_label"], dtype=np.int64), np.asarray(L["pairs"], dtype=np.int64).reshape(-1, 2)[:, 0]]
        if r + 1 < world:
            mine.append(np.asarray(lists[r + 1]["pairs"], dtype=np.int64).reshape(-1, 2)[:, 1])
        u = np.unique(np.concatenate(mine))
        uniq.append(u)
        base.append(base[-1] + len(u))
    n_nodes = base[-1]

    def node(r, labels):
        return base[r] + np.searchsorted(uniq[r], np.asarray(labels, dtype=np.int64))
    pa, pb = [], []
    for r, L in enumerate(lists):
        p = np.asarray(L["pairs"], dtype=np.int64).reshape(-1, 2)
        if len(p):
            assert r > 0, "rank 0 has no lower neighbour"
            pa.append(node(r, p[:, 0]))
            pb.append(node(r - 1, p[:, 1]))
    pa = np.concatenate(pa) if pa else np.zeros(0, np.int64)
    pb = np.concatenate(pb) if pb else np.zeros(0, np.int64)
    g = coo_matrix((np.ones(len(pa), dtype=np.int8), (pa, pb)), shape=(n_nodes, n_nodes))
    n_comp, comp = connected_components(g, directed=False) if n_nodes else (0, np.zeros(0, np.int64))
    cn, cx, cv, cnx, cneg = [], [], [], [], []
    for r, L in enumerate(lists):
        has = np.asarray(L["cand_has"]).astype(bool)
        cn.append(node(r, np.asarray(L["cand_label"])[has]))
        cx.append(np.asarray(L["cand_x"], dtype=np.float64)[has])
        cv.append(np.asarray(L["cand_vertex_key"], dtype=np.int64)[has])
        cnx.append(np.asarray(L["cand_nx"], dtype=np.float64)[has])
        cneg.append(np.asarray(L["cand_negative"], dtype=np.int64)[has])
    cn, cx, cv, cnx, cneg = [np.concatenate(a) if a else np.zeros(0) for a in (cn, cx, cv, cnx, cneg)]
    flip_of_comp = np.zeros(n_comp, dtype=np.uint8)
    decided = np.zeros(n_comp, dtype=bool)
    if len(cn):
        cc = comp[cn.astype(np.int64)]
        order = np.lexsort((1 - cneg, cnx, cv, cx, cc))      # per component: the last one has the largest (x, edge id, |normal_x|)
        last = np.ones(len(order), dtype=bool)
        last[:-1] = cc[order][1:] != cc[order][:-1]
        win = order[last]
        flip_of_comp[cc[win]] = cneg[win].astype(np.uint8)
        decided[cc[win]] = True
    out = []
    for r in range(world):
        labels = uniq[r].astype(np.uint32)
        c = comp[base[r]:base[r + 1]]
        keep = decided[c]
        out.append((labels[keep], flip_of_comp[c][keep]))
    return out, dict(nodes=int(n_nodes), components=int(n_comp), pairs=int(len(pa)),
                     unmatched=int(sum(int(L.get("unmatched", 0)) for L in lists)))


def shard_layout(n0, world, rank, layers=None):
    """planes and cells of rank `rank`'s local array for the sharded Level 1 -> dict(i0, i1 own planes [i0, i1), e0, e1 local
    planes [e0, e1), own_lo, own_hi own cell layers of the local array)"""
    H = SHARD_LAYERS if layers is None else layers
    i0, i1 = slab_bounds(n0, world, rank)
    e0 = max(0, i0 - H) if rank > 0 else i0
    e1 = min(n0, i1 + H + 1) if rank + 1 < world else i1
    own_hi_plane = i1 if rank + 1 < world else i1 - 1          # cells [i0, own_hi_plane)
    return dict(i0=i0, i1=i1, e0=e0, e1=e1, own_lo=i0 - e0, own_hi=own_hi_plane - e0)


def shard_local(ctx, local, layout, value, global_shape, clean=True, torch_device=None):
    """first half on one rank: march the local array (own planes + neighbours' layers) and run the local post-pass.
    -> dict: the start-triangle candidates (edge ids already those of the whole volume), own1 = (hash, label) of the own
    triangles next to the lower neighbour, copy4 = (hash, label) of the copies of the upper neighbour's first layer (torch
    tensors on torch_device, else numpy), counts"""
    from . import _ffi
    n0, n1, n2 = [int(n) for n in global_shape]
    ctx.set_origin(layout["e0"], 0, 0)
    if type(local).__module__.split(".")[0] == "torch":
        if local.is_cuda:
            ctx.adopt_device_grid(local.data_ptr(), tuple(local.shape), keepalive=local)
        else:
            ctx.upload_grid(local.numpy())
    else:
        ctx.upload_grid(local)
    ctx.set_reference_corner((n0 - 1, n1 - 1, n2 - 1))
    try:
        ctx.extract3d(float(value), _ffi.CX_DIAG_CPYTHON310)
        L = ctx.shard_begin(layout["own_lo"], layout["own_hi"], 0 if clean else 1)
    finally:
        ctx.set_reference_corner((0, 0, 0))
    off = (np.int64(layout["e0"]) * n1 * n2) << 3
    L["cand_vertex_key"] = L["cand_vertex_key"].astype(np.int64) + off
    L["key_offset"] = off
    L["own1"] = ctx.shard_boundary(1, L["n_own_lower"], torch_device)
    L["copy4"] = ctx.shard_boundary(4, L["n_upper_copies"], torch_device)
    return L


def shard_small(L, pairs, unmatched):
    "what goes to rank 0: the label pairs with the lower neighbour and the candidates (no per-triangle data)"
    out = {k: L[k] for k in ("cand_label", "cand_x", "cand_vertex_key", "cand_nx", "cand_negative", "cand_has")}
    out["pairs"] = pairs
    out["unmatched"] = int(unmatched)
    return out


def shard_finish(ctx, L, answer, download=True):
    "second half on one rank: the agreed flips -> dict(counts, points, triangles, keys) of the rank's own part"
    counts = ctx.shard_finish(answer[0], answer[1])
    counts.update(L["counts"])
    out = dict(counts=counts)
    if download:
        pts, tris = ctx.download_level1(counts)
        out.update(points=pts, triangles=tris, keys=ctx.download_level1_keys(counts).astype(np.int64) + L["key_offset"])
    return out


def exchange_boundary_lists(L, rank, world, dist, device):
    """rank r sends its copies of rank r+1's first layer (hash, label) up and receives rank r-1's copies of its own: tensors on
    `device` (RCCL with backend "nccl"; staged through the host for gloo).  -> (hash, label) of the lower neighbour's list, or
    None on rank 0"""
    import torch
    staged = dist.get_backend() == "gloo"
    wire = torch.device("cpu") if staged else device
    h4, l4 = L["copy4"]
    n_up = torch.tensor([int(h4.numel())], dtype=torch.int64, device=wire)
    n_low = torch.zeros(1, dtype=torch.int64, device=wire)
    ops = []
    if rank + 1 < world:
        ops.append(dist.P2POp(dist.isend, n_up, rank + 1))
    if rank > 0:
        ops.append(dist.P2POp(dist.irecv, n_low, rank - 1))
    for w in dist.batch_isend_irecv(ops):
        w.wait()
    got = None
    ops = []
    keep = []
    if rank + 1 < world and int(h4.numel()):
        sh, sl = h4.to(wire).contiguous(), l4.to(wire).contiguous()
        keep += [sh, sl]
        ops += [dist.P2POp(dist.isend, sh, rank + 1), dist.P2POp(dist.isend, sl, rank + 1)]
    if rank > 0:
        n = int(n_low.item())
        rh = torch.empty(n, dtype=torch.int64, device=wire)
        rl = torch.empty(n, dtype=torch.int32, device=wire)
        if n:
            ops += [dist.P2POp(dist.irecv, rh, rank - 1), dist.P2POp(dist.irecv, rl, rank - 1)]
        got = (rh, rl)
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if got is not None:
        got = (got[0].to(device), got[1].to(device))
    return got


def level1_slabs_sharded(own_planes, value, rank, world, global_shape, device=0, clean=True, dist=None, context=None, download=True,
                         object_group=None):
    """Level 1 of a volume spread over the ranks in slabs along axis 0, WITHOUT gathering the mesh: every rank returns its own
    part -- dict(points (V,3) float64 in the coordinates of the whole volume, triangles (T,3) int32 into them, wound as the
    reference winds the whole surface, keys (V,) int64 edge id of every vertex in the whole volume (vertices next to a slab
    boundary appear on both sides with the same id and coordinates), counts, stats (rank 0), ms, boundary).  The union over the
    ranks is the Level-1 mesh of the undivided volume (assemble_level1).  Per-triangle data only travels between neighbours
    (12 bytes per boundary triangle, device to device with RCCL); rank 0 sees label pairs and candidates.
    object_group: process group for the two small object collectives (gather_object / scatter_object_list), e.g. a gloo group
    with a timeout beside an RCCL job; default: the default group."""
    import time
    import torch
    from . import _ffi
    if dist is None and world > 1:
        import torch.distributed as dist
    n0 = int(global_shape[0])
    lay = shard_layout(n0, world, rank)
    t = own_planes if isinstance(own_planes, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(own_planes, dtype=np.float32))
    assert t.shape[0] == lay["i1"] - lay["i0"]
    # every rank checks EVERY rank's slab (slab_bounds is the same arithmetic everywhere): a slab thinner than the layers it has to
    # hand to its neighbours raises on all ranks alike, before anybody waits in a collective
    if world > 1:
        thin = [r for r in range(world) if slab_bounds(n0, world, r)[1] - slab_bounds(n0, world, r)[0] < SHARD_LAYERS + 1]
        if thin:
            raise ValueError("sharded Level 1: %d planes over %d ranks leaves rank(s) %s fewer than the %d planes a slab hands to its neighbours"
                             % (n0, world, thin, SHARD_LAYERS + 1))
    t0 = time.perf_counter()
    local, nb = exchange_planes(t, rank, world, SHARD_LAYERS, SHARD_LAYERS + 1, dist)
    assert nb == lay["own_lo"] and local.shape[0] == lay["e1"] - lay["e0"], (nb, lay, tuple(local.shape))
    if local.is_cuda:
        torch.cuda.synchronize(local.device)
    ctx = context or _ffi.Context(device)
    gpu = torch.device("cuda", ctx.device if hasattr(ctx, "device") else device)
    t1 = time.perf_counter()
    # a rank whose local part fails must not leave the others waiting in the exchange: everybody learns of it first
    err = None
    try:
        L = shard_local(ctx, local, lay, value, global_shape, clean, torch_device=gpu)
    except Exception as e:      # noqa: BLE001 -- re-raised below, on every rank
        err = e
    if world > 1:
        on_device = dist.get_backend() == "nccl"
        flag = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device=gpu if on_device else torch.device("cpu"))
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            raise RuntimeError("sharded Level 1: the local part failed on some rank" + ("" if err is None else " (this one): %s" % err))
    elif err is not None:
        raise err
    t2 = time.perf_counter()
    stats = None
    if world == 1:
        mine = (np.zeros(0, np.uint32), np.zeros(0, np.uint8))
    else:
        # Everything between here and the scatter can fail on ONE rank (the neighbour exchange, the pairing, rank 0's merge) while
        # the others sit in a collective.  So nobody raises in between: an error travels WITH the small object every rank sends to
        # rank 0, rank 0 adds its own (and `unmatched` boundary triangles: components that are one surface would stay apart and the
        # slabs' windings could disagree), and the verdict comes back to every rank in place of its flips.
        small, err = None, None
        try:
            lower = exchange_boundary_lists(L, rank, world, dist, gpu)
            if lower is not None:
                pairs, unmatched = pair_labels(L["own1"][0], L["own1"][1], lower[0], lower[1])
            else:
                pairs, unmatched = np.zeros((0, 2), dtype=np.int64), 0
            small = shard_small(L, pairs, unmatched)
        except Exception as e:      # noqa: BLE001 -- carried to rank 0, raised below on every rank
            err = "%s: %s" % (type(e).__name__, e)
        gathered = [None] * world if rank == 0 else None
        dist.gather_object({"error": err} if err is not None else small, gathered, dst=0, group=object_group)
        answers = [None] * world
        if rank == 0:
            failed = ["rank %d: %s" % (r, g["error"]) for r, g in enumerate(gathered) if g is not None and g.get("error")]
            if not failed:
                try:
                    n_unmatched = sum(int(g.get("unmatched", 0)) for g in gathered)
                    if n_unmatched:
                        failed = ["%d boundary triangles are known to one slab only (or share a hash): the slabs' components cannot be united" % n_unmatched]
                    else:
                        answers, stats = merge_shard_components(gathered)
                except Exception as e:      # noqa: BLE001
                    failed = ["rank 0 (merge): %s: %s" % (type(e).__name__, e)]
            if failed:
                answers = [{"error": "; ".join(failed)}] * world
        box = [None]
        dist.scatter_object_list(box, answers if rank == 0 else None, src=0, group=object_group)
        mine = box[0]
        if isinstance(mine, dict) and mine.get("error"):
            raise RuntimeError("sharded Level 1 failed (raised on every rank): " + mine["error"])
    t3 = time.perf_counter()
    out = shard_finish(ctx, L, mine, download)
    t4 = time.perf_counter()
    out["stats"] = stats
    out["ms"] = dict(halo=(t1 - t0) * 1e3, local=(t2 - t1) * 1e3, exchange=(t3 - t2) * 1e3, finish=(t4 - t3) * 1e3)
    out["boundary"] = dict(triangles=int(L["n_own_lower"] + L["n_upper_copies"]), components=int(len(L["cand_label"])))
    return out


def assemble_level1(parts):
    """parts: [(keys (V,) int64, points (V,3), triangles (T,3))] of all ranks -> (keys sorted unique, points, triangles): one mesh;
    a vertex that two slabs hold must have the same coordinates in both"""
    keys = np.concatenate([np.asarray(p[0], dtype=np.int64) for p in parts])
    pts = np.concatenate([np.asarray(p[1], dtype=np.float64).reshape(-1, 3) for p in parts])
    ukeys, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    if not np.array_equal(pts[first][inv], pts):
        raise RuntimeError("a vertex has different coordinates in two slabs")
    tris, at = [], 0
    for p in parts:
        tris.append(inv[at + np.asarray(p[2], dtype=np.int64).reshape(-1, 3)])
        at += len(p[0])
    return ukeys, pts[first], np.concatenate(tris) if tris else np.zeros((0, 3), np.int64)
