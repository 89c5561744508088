"""Frame-sharded encoding of ONE long video over the GPUs of a node; one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Rank r owns the contiguous frames [lo, hi) of the frames that survive the a1 cap.  Cross-rank data flow per video:
  0. a1 (frame budget + uniform sub-sampling) is host integer logic on (T0, text length): every rank computes the same
     selection (`frame_plan`) and loads / is handed only its own frames.
  1. DINO on the local frames; every rank but the first sends the feature rows of its FIRST frame (576 x 1536 16-bit values,
     1.77 MB) to its left neighbour, which needs them for its last adjacent similarity (round 1 re-encoded that frame on
     both ranks: one frame in 65 sounds cheap, but it pushes the DINOv2 qkv / fc1 GEMMs of a 64-frame shard over a tile-round
     boundary of the persistent kernel - ~2 % of the step against ~0.1 ms of xGMI transfer); callers that still pass the
     halo frame's pixels get the old behaviour.  All-gather of the T-1 fp32 similarities -> every rank runs the identical
     stable selection -> identical plan.
  2. SigLIP + connector on the local frames (no communication).  Audio (a20): a rank runs BEATs on the 10-second windows
     its own seconds fall into (a window that straddles a rank boundary is encoded by both ranks, like the DINO halo)
     and builds the audio tokens of its own frames; no exchange.
  3. Chunks whose key frame lives on another rank: the owner sends its K x Dq query block (K=16: 24 KB) point to point
     (query_type 'learned': every rank holds the same `query_tokens`, nothing is sent).
  4. Q-Former on the local compressed frames; every rank emits the tokens of its own frames in plan order;
     all-gather (padded to the longest shard; shard lengths are known from the plan) -> identical full stream on every
     rank, tail clipping (a19) already applied through the shared plan.
The numerical work is delegated to an engine object (pipeline.VideoEncoder on GPUs; a test double in the gloo tests);
the transport to a comm object (TorchComm = torch.distributed; the tests also plug in an in-process thread transport to
rehearse world sizes a one-GPU box cannot host as processes).
"""
import numpy as np
import torch

from . import segment as seg
from .pipeline import sample_indicator


def owner_of(frame, ranges):
    for r, (lo, hi) in enumerate(ranges):
        if lo <= frame < hi:
            return r
    raise ValueError(frame)


def split_plan(plan, ranges, Nf, K):
    """Per rank: emission pairs (table, row) in plan order with LOCAL rows (int32 array [n_r, 2]); the separator follows
    its frame.  Also returns, per rank, the dict global compressed-frame index -> local index."""
    starts = np.asarray([lo for lo, _ in ranges], dtype=np.int64)
    comp_frames = np.asarray(plan["comp_frames"] if plan["comp_frames"] else [0], dtype=np.int64)
    n_comp = len(plan["comp_frames"])
    comp_owner = np.searchsorted(starts, comp_frames[:max(n_comp, 1)], side="right") - 1
    comp_local_idx = np.zeros(max(n_comp, 1), dtype=np.int64)
    comp_local = [dict() for _ in ranges]
    for r in range(len(ranges)):
        m = np.nonzero(comp_owner[:n_comp] == r)[0]
        comp_local_idx[m] = np.arange(len(m))
        comp_local[r] = {int(g): i for i, g in enumerate(m.tolist())}
    kind, a, b = plan.kind, plan.a, plan.b
    n = len(kind)
    a_c = np.where(kind == 1, a, 0)
    frame = np.where(kind == 0, a, np.where(kind == 1, comp_frames[a_c], -1))
    last = np.maximum.accumulate(np.where(kind != 2, np.arange(n), -1)) if n else np.zeros(0, dtype=np.int64)
    frame_ff = np.where(last >= 0, frame[np.maximum(last, 0)], starts[0]) if n else frame
    owner = np.searchsorted(starts, frame_ff, side="right") - 1
    row = np.where(kind == 0, (a - starts[owner]) * Nf + b, np.where(kind == 1, comp_local_idx[a_c] * K + b, 0))
    both = np.stack([kind, row], 1).astype(np.int32)
    per = [np.ascontiguousarray(both[owner == r]) for r in range(len(ranges))]
    return per, comp_local


class TorchComm:
    """torch.distributed transport (RCCL on the GPUs of a node; gloo in the CPU tests and in the one-GPU rehearsals).
    RCCL operations are ordered on the stream; gloo's point-to-point operations read a device tensor from the host through
    its raw pointer with no stream ordering at all, so with gloo the device is synchronised before anything is posted."""

    def __init__(self, rank, world, group=None):
        self.rank, self.world, self.group = rank, world, group
        self._host_side = None

    def _sync_if_host_side(self, t):
        if self._host_side is None:
            import torch.distributed as dist
            self._host_side = dist.get_backend(self.group) != "nccl"
        if self._host_side and t is not None and t.is_cuda:
            torch.cuda.synchronize(t.device)

    def _host_side_now(self):
        if self._host_side is None:
            import torch.distributed as dist
            self._host_side = dist.get_backend(self.group) != "nccl"
        return self._host_side

    def all_gather(self, t):
        import torch.distributed as dist
        self._sync_if_host_side(t)
        out = [torch.empty_like(t) for _ in range(self.world)]
        dist.all_gather(out, t, group=self.group)
        return out

    def all_gather_into(self, out, t):
        """out [world * n, ...] <- the ranks' t [n, ...] in rank order: ONE collective into a caller-owned buffer (RCCL: the
        direct all-gather over the xGMI links, no list of per-rank outputs, no concatenation afterwards)."""
        import torch.distributed as dist
        self._sync_if_host_side(t)
        dist.all_gather_into_tensor(out, t, group=self.group)
        self._sync_if_host_side(t)

    def exchange(self, sends, recvs):
        """sends: [(tensor, dst)], recvs: [(buffer, src)] - posted in this order on every rank, completed before return."""
        import torch.distributed as dist
        ops_ = [dist.P2POp(dist.isend, t, dst, self.group) for t, dst in sends] + \
               [dist.P2POp(dist.irecv, buf, src, self.group) for buf, src in recvs]
        if ops_:
            self._sync_if_host_side((sends + recvs)[0][0])
            for rq in dist.batch_isend_irecv(ops_):
                rq.wait()
            self._sync_if_host_side((sends + recvs)[0][0])


class ShardedVideoEncoder:
    def __init__(self, engine, rank, world, group=None, comm=None):
        self.e, self.rank, self.world = engine, rank, world
        self.comm = comm if comm is not None else TorchComm(rank, world, group)
        self._bufs = {}        # (tag, rows, cols, dtype) -> preallocated send / receive buffers of the token all-gather
        self._maps = {}        # shard lengths -> device index map of the compaction gather

    # ---- a1 on every rank --------------------------------------------------------------------------------------------
    def frame_plan(self, T0, budget_text_len, frame_cap=224, video_index=None, halo=False):
        """tdc/cambrian_arch.py:899-935 for a T0-frame video: which input frames survive the budget, which of them this rank
        encodes (`siglip_frames`; `dino_frames` = the same - plus the next rank's first frame with halo=True, the
        re-encode-instead-of-exchange variant), and the per-second 0/1 vector the audio interleave needs.  Pure host
        integers: identical on every rank."""
        cfg = self.e.cfg
        idx = seg.uniform_indices(T0, min(seg.get_max_num_frames(budget_text_len, cfg), frame_cap))
        T = len(idx)
        lo, hi = seg.shard_ranges(T, self.world)[self.rank]
        want_halo = bool(halo)
        halo = 1 if (halo and lo < hi < T) else 0    # a rank without frames (T < world) has no halo either
        return dict(idx=idx, T=T, lo=lo, hi=hi, siglip_frames=idx[lo:hi], dino_frames=idx[lo:hi + halo], recompute_halo=want_halo,
                    sample_indices=sample_indicator(T0, idx, video_index))

    def _tokens_per_frame(self, image_size, audio):
        """rows per frame in the Q-Former KV / static emission (cur_h * (cur_w + 1), + 50 with audio) without running the
        connector - only a rank that owns no frame needs it (to lay out the shared plan)."""
        side = getattr(self.e, "side", None)
        if side is None:
            return self.e.N + (0 if audio is None else 2)          # CPU test double
        r0, r1, c0, c1 = seg.unpad_bounds(side, side, image_size)
        return (r1 - r0) * (c1 - c0 + 1) + (50 if audio is not None else 0)

    def _all_gather_var(self, t, counts):
        """all-gather of 1-D/2-D tensors with per-rank leading sizes `counts` (known to every rank): the small exchanges
        (similarities) and engines / transports without the buffered form."""
        mx = max(counts)
        pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        if t.shape[0]:
            pad[: t.shape[0]] = t
        out = self.comm.all_gather(pad)
        return torch.cat([o[:c] for o, c in zip(out, counts)], 0)

    def _buf(self, tag, rows, cols, dtype, device):
        key = (tag, rows, cols, dtype, str(device))
        b = self._bufs.get(key)
        if b is None:
            for k in [k for k in self._bufs if k[0] == tag]:      # one buffer per role: a new shape replaces the old one
                del self._bufs[k]
            b = self._bufs[key] = torch.zeros(rows, cols, dtype=dtype, device=device)
        return b

    def _gather_tokens(self, emit_local, counts, cols, dtype, device):
        """The final exchange: every rank's emitted rows -> the whole stream on every rank.
        `emit_local(out)` writes this rank's counts[rank] rows into out[:counts[rank]].  Buffered form (engine with
        `compact_rows`, transport with `all_gather_into`): the rank emits straight into its padded send buffer, ONE
        all_gather_into_tensor fills a preallocated [world * max, cols] buffer and one row gather (tdc_gather_rows with an
        index map cached per shard-length tuple) drops the padding - no per-step allocation besides the result, no list of
        per-rank tensors, no torch.cat.  At the bench's size (T = 512, K = 144, world 8) a rank contributes ~9.4 k rows =
        67 MB and receives 538 MB."""
        mx, n = max(counts), counts[self.rank]
        e = self.e
        if not (hasattr(e, "compact_rows") and hasattr(self.comm, "all_gather_into")):
            local = torch.zeros(0, cols, dtype=dtype, device=device)
            if n:
                local = torch.empty(n, cols, dtype=dtype, device=device)
                emit_local(local)
            return self._all_gather_var(local, counts)
        send = self._buf("send", mx, cols, dtype, device)
        recv = self._buf("recv", self.world * mx, cols, dtype, device)
        if n:
            emit_local(send)
        self.comm.all_gather_into(recv, send)
        key = tuple(counts)
        idx = self._maps.get(key)
        if idx is None:
            self._maps.clear()
            rows = np.concatenate([r * mx + np.arange(c, dtype=np.int64) for r, c in enumerate(counts)]) if sum(counts) else \
                np.zeros(0, dtype=np.int64)
            pairs = np.zeros((len(rows), 2), dtype=np.int32)
            pairs[:, 1] = rows
            idx = self._maps[key] = torch.from_numpy(pairs).to(device)
        return e.compact_rows(recv, idx, cols)

    def _refine_selection(self, band, sims, mns, eps, px_dino_local_halo, ranges, recompute_halo):
        """a5 at the reference's precision, sharded (pipeline.encode_video_with does the same serially): the pairs of `band` - the
        same list on every rank - are re-encoded by the engine's precise DINOv2 tower.  Pair i belongs to the rank that owns frame
        i; it needs frame i + 1 as well, which for the last local frame is the right neighbour's first: that rank sends the precise
        feature rows of its first frame (as the boundary exchange of step 1 does for the fast tower), unless the caller handed
        this rank the halo frame's pixels (recompute_halo).  The refined values are all-gathered in band order; every rank then
        runs the same host selection."""
        e, rank, world = self.e, self.rank, self.world
        lo, hi = ranges[rank]
        inband = set(band)
        mine = [i for i in band if lo <= i < hi]
        frames = {f for i in mine for f in (i, i + 1) if f < hi}
        send_first = rank > 0 and (lo - 1) in inband and hi > lo and not recompute_halo      # left neighbour's last pair
        if send_first:
            frames.add(lo)
        cross = rank < world - 1 and hi > lo and (hi - 1) in inband                          # my last pair crosses the boundary
        if cross and recompute_halo:
            frames.add(hi)                                                                   # its pixels are local (row hi - lo)
        frames = sorted(frames)
        feats = {}
        if frames:
            idx = torch.tensor([f - lo for f in frames], device=px_dino_local_halo.device)
            fp = e.precise_dino(px_dino_local_halo[idx])
            Pp = fp.shape[0] // len(frames)
            feats = {f: fp[j * Pp:(j + 1) * Pp] for j, f in enumerate(frames)}
        if not recompute_halo:
            sends = [(feats[lo].contiguous(), rank - 1)] if send_first else []
            recvs = []
            if cross:
                ref_rows = next(iter(feats.values()))
                buf = torch.empty_like(ref_rows)
                feats[hi] = buf
                recvs = [(buf, rank + 1)]
            self.comm.exchange(sends, recvs)
        dev = px_dino_local_halo.device
        local = e.pair_sims(feats, [(i, i + 1) for i in mine]) if mine else torch.zeros(0, dtype=torch.float32, device=dev)
        counts = [sum(1 for i in band if l <= i < h) for (l, h) in ranges]
        refined = self._all_gather_var(local.to(torch.float32), counts).tolist()
        return seg.select_refined(sims, mns, eps, band, refined)

    def encode_video(self, px_siglip_local, px_dino_local_halo, T, image_size, n_text_tokens, prompt_ids, audio=None,
                     sample_indices=None, recompute_halo=False):
        """px_siglip_local: frames [lo,hi) of the T frames that survive a1 (frame_plan); px_dino_local_halo: the same frames for
        the DINOv2 tower (the boundary frame's features are then exchanged), or frames [lo, hi + 1) - the next rank's first
        frame appended (none on the last rank) - with recompute_halo=True (on EVERY rank), to re-encode that frame here instead.  audio: as in the serial path - a full [T, 50, 768] token tensor, or a dict
        with "audio_tokens" / "beats_windows" (all windows) / "audio_wav" (raw 16 kHz waveform: BEATs runs here, on this
        rank's windows only); sample_indices: frame_plan's per-second vector (all ones when a1 did not cap)."""
        e, rank, world = self.e, self.rank, self.world
        ranges = seg.shard_ranges(T, world)
        lo, hi = ranges[rank]
        Tl = hi - lo
        cfg = e.cfg
        # 1. DINO (+halo) -> similarities -> identical segmentation everywhere
        n_d = px_dino_local_halo.shape[0]
        assert n_d == Tl + (1 if (recompute_halo and lo < hi < T) else 0) and px_siglip_local.shape[0] == Tl
        if Tl == 0:
            # more ranks than frames (T < world <= 8 < max_num_segments + 1: the pass-through case, no similarities, no
            # Q-Former): this rank only takes part in the final all-gather
            assert T <= cfg.get("max_num_segments", 24) + 1
            Nf = self._tokens_per_frame(image_size, audio)
            plan = seg.emit_plan(T, Nf, e.K, list(range(T)), cfg["tokenizer_model_max_length"] -
                                 cfg.get("inference_max_length", 16) - n_text_tokens, cfg.get("add_static", True))
            pairs, _ = split_plan(plan, ranges, Nf, e.K)
            if cfg.get("query_type", "Avg_pool") != "learned":
                self.comm.exchange([], [])
            return self._gather_tokens(lambda out: None, [len(p) for p in pairs], e.H, e.dtype, px_siglip_local.device)
        # engine.two_streams (small shards: bench.py switches it on at <= 128 frames per rank): the SigLIP tower runs on a side
        # stream beside the DINOv2 tower, so the partly filled last tile rounds of one tower's GEMMs are filled by the other's
        # workgroups (T = 64 on one GPU: +0.6-1.1 %; at 512 frames per launch the tails are too short to matter)
        side, sig_early = None, None
        if getattr(e, "two_streams", False) and px_siglip_local.is_cuda:
            side = e.tower_stream() if hasattr(e, "tower_stream") else torch.cuda.Stream(device=px_siglip_local.device)
            side.wait_stream(torch.cuda.current_stream(px_siglip_local.device))
            with torch.cuda.stream(side):
                sig_early = e.tower("siglip", px_siglip_local)
        dino_all = e.tower("dino", px_dino_local_halo)
        P = dino_all.shape[0] // n_d
        mns = cfg.get("max_num_segments", 24)
        sig = None
        if T <= mns + 1:
            seg_idx = list(range(T))
        else:
            # T > 25 >= 3 * world: every rank owns frames.  Local pairs, then the pair across the right-hand boundary.
            sims_parts = []
            if not recompute_halo:
                first = dino_all[:P].contiguous()
                halo_rows = torch.empty_like(first) if rank < world - 1 else None
                self.comm.exchange([(first, rank - 1)] if rank > 0 else [],
                                   [(halo_rows, rank + 1)] if rank < world - 1 else [])
                if Tl >= 2:
                    sims_parts.append(e.sims_tensor(dino_all, Tl))
                if halo_rows is not None:
                    sims_parts.append(e.sims_tensor(torch.cat([dino_all[(Tl - 1) * P: Tl * P], halo_rows], 0), 2))
            elif n_d >= 2:
                sims_parts.append(e.sims_tensor(dino_all, n_d))
            sims_local = torch.cat(sims_parts, 0) if sims_parts else \
                torch.zeros(0, dtype=torch.float32, device=dino_all.device)
            # the local SigLIP tower is enqueued before the exchange: the device works while the similarities travel - on a
            # side stream that waits only for the similarity kernels (engine.mark / after), so neither the collective nor
            # the host read queues behind the tower (with a host-side transport - gloo - the device is synchronised anyway)
            ev = e.mark() if (hasattr(e, "mark") and sims_local.is_cuda and not getattr(self.comm, "_host_side_now", lambda: True)()) \
                else None
            if side is None:
                sig = e.tower("siglip", px_siglip_local)
            counts = [(h - l) - (0 if r < world - 1 else 1) for r, (l, h) in enumerate(ranges)]
            if ev is not None:
                with e.after(ev):
                    sims_local.record_stream(torch.cuda.current_stream(sims_local.device))
                    sims = self._all_gather_var(sims_local, counts).tolist()
            else:
                sims = self._all_gather_var(sims_local, counts).tolist()
            assert len(sims) == T - 1
            seg_idx = seg.select_segments(sims, mns)
            eps = getattr(e, "selection_eps", None)
            band = seg.selection_band(sims, mns, eps) if eps else []         # host integers: identical on every rank
            if band and seg.band_allowed(band, T, getattr(e, "selection_max_fraction", 0.125)):
                seg_idx = self._refine_selection(band, sims, mns, eps, px_dino_local_halo, ranges, recompute_halo)
        dino = dino_all[: Tl * P]
        # 2. local towers + connector (+ audio rows of the local frames)
        if side is not None:
            torch.cuda.current_stream(px_siglip_local.device).wait_stream(side)
            sig = sig_early
            sig.record_stream(torch.cuda.current_stream(px_siglip_local.device))
        elif sig is None:
            sig = e.tower("siglip", px_siglip_local)
        X, sizes = e.connector(sig, dino, Tl, [tuple(image_size)] * Tl)
        N = X.shape[0] // Tl
        K = e.K
        if sample_indices is None:
            sample_indices = [1] * T
        Xf, Nf = e.with_audio(X, Tl, N, e.local_audio(audio, sample_indices, T, lo, hi))
        # 3. shared plan, query hand-off
        max_visual_len = cfg["tokenizer_model_max_length"] - cfg.get("inference_max_length", 16) - n_text_tokens
        plan = seg.emit_plan(T, Nf, K, seg_idx, max_visual_len, cfg.get("add_static", True))
        pairs, comp_local = split_plan(plan, ranges, Nf, K)
        keys = plan["key_frames"]
        my_comp = [gi for gi, f in enumerate(plan["comp_frames"]) if lo <= f < hi]
        pid = prompt_ids if cfg.get("text_input", True) else None
        comp = None
        if cfg.get("query_type", "Avg_pool") == "learned":          # cambrian_arch.py:1639-1640: one shared query block
            if my_comp:
                comp = e.compress_frames(Xf, Nf, [plan["comp_frames"][gi] - lo for gi in my_comp], e.learned_queries(),
                                         [0] * len(my_comp), pid)
        else:
            need = sorted(set(plan["comp_chunk"][gi] for gi in my_comp))
            owned = [c for c, s in enumerate(keys) if lo <= s < hi]
            q_owned = e.make_queries(Xf, N, Nf, [keys[c] - lo for c in owned]) if owned else None
            users = {}                               # chunk -> ranks that hold compressed frames of it
            for gi, f in enumerate(plan["comp_frames"]):
                users.setdefault(plan["comp_chunk"][gi], set()).add(owner_of(f, ranges))
            sends, recvs, recv_buf = [], [], {}
            for ci, c in enumerate(owned):                   # ascending chunk id on both sides: pairwise order matches
                for r in sorted(users.get(c, ())):
                    if r != rank:
                        sends.append((q_owned[ci * K:(ci + 1) * K].contiguous(), r))
            for c in need:
                if not (lo <= keys[c] < hi):
                    buf = torch.empty((K, e.query_width()), dtype=e.dtype, device=X.device)
                    recv_buf[c] = buf
                    recvs.append((buf, owner_of(keys[c], ranges)))
            self.comm.exchange(sends, recvs)
            if my_comp:
                blocks = []
                for c in need:
                    if c in recv_buf:
                        blocks.append(recv_buf[c])
                    else:
                        ci = owned.index(c)
                        blocks.append(q_owned[ci * K:(ci + 1) * K])
                qtable = torch.cat(blocks, 0)
                qsrc = [need.index(plan["comp_chunk"][gi]) for gi in my_comp]
                comp = e.compress_frames(Xf, Nf, [plan["comp_frames"][gi] - lo for gi in my_comp], qtable, qsrc, pid)
        # 4. local emission + all-gather
        mine = pairs[rank]

        def emit_local(out):
            if getattr(e, "emit_into", None) is not None:
                e.emit_into(Xf, comp, mine, out)              # the a19 gather writes the send buffer directly
            else:
                out[: len(mine)].copy_(e.emit(Xf, comp, mine))
        return self._gather_tokens(emit_local, [len(p) for p in pairs], e.H, e.dtype, X.device)
