"""Frame-sharded encoding of ONE long video over the GPUs of a node; one process per GPU,
torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

Rank r owns the contiguous frames [lo, hi).  Cross-rank data flow per video:
  1. DINO on the local frames plus a one-frame halo (the next rank's first frame is re-encoded locally: 1/(T/P) extra
     tower work instead of a 1.7 MB send that would serialise the ranks) -> local adjacent similarities;
     all-gather of the T-1 fp32 similarities -> every rank runs the identical stable selection -> identical plan.
  2. SigLIP + connector on the local frames (no communication).
  3. Chunks whose key frame lives on another rank: the owner sends its K x Dq query block (K=16: 24 KB) point to point.
  4. Q-Former on the local compressed frames; every rank emits the tokens of its own frames in plan order;
     all-gather (padded to the longest shard; shard lengths are known from the plan) -> identical full stream on every
     rank, tail clipping (a19) already applied through the shared plan.
The numerical work is delegated to an engine object (pipeline.VideoEncoder on GPUs; a test double in the gloo tests).
"""
import torch
import torch.distributed as dist

from . import segment as seg


def owner_of(frame, ranges):
    for r, (lo, hi) in enumerate(ranges):
        if lo <= frame < hi:
            return r
    raise ValueError(frame)


def split_plan(plan, ranges, Nf, K):
    """Per rank: emission pairs (table,row) in plan order with LOCAL rows; the separator follows its frame."""
    per = [[] for _ in ranges]
    comp_local = [dict() for _ in ranges]       # global comp idx -> local comp idx
    for gi, f in enumerate(plan["comp_frames"]):
        r = owner_of(f, ranges)
        comp_local[r][gi] = len(comp_local[r])
    cur = 0
    for e in plan["src"]:
        if e[0] == "f":
            cur = owner_of(e[1], ranges)
            per[cur].append((0, (e[1] - ranges[cur][0]) * Nf + e[2]))
        elif e[0] == "c":
            cur = owner_of(plan["comp_frames"][e[1]], ranges)
            per[cur].append((1, comp_local[cur][e[1]] * K + e[2]))
        else:
            per[cur].append((2, 0))
    return per, comp_local


class ShardedVideoEncoder:
    def __init__(self, engine, rank, world, group=None):
        self.e, self.rank, self.world, self.group = engine, rank, world, group

    def _all_gather_var(self, t, counts):
        """all-gather of 1-D/2-D tensors with per-rank leading sizes `counts` (known to every rank)."""
        mx = max(counts)
        pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        if t.shape[0]:
            pad[: t.shape[0]] = t
        out = [torch.empty_like(pad) for _ in range(self.world)]
        dist.all_gather(out, pad, group=self.group)
        return torch.cat([o[:c] for o, c in zip(out, counts)], 0)

    def encode_video(self, px_siglip_local, px_dino_local_halo, T, image_size, n_text_tokens, prompt_ids):
        """px_siglip_local: frames [lo,hi); px_dino_local_halo: frames [lo, hi + 1) (no halo on the last rank)."""
        e, rank, world = self.e, self.rank, self.world
        ranges = seg.shard_ranges(T, world)
        lo, hi = ranges[rank]
        Tl = hi - lo
        cfg = e.cfg
        # 1. DINO (+halo) -> similarities -> identical segmentation everywhere
        n_d = px_dino_local_halo.shape[0]
        assert n_d == Tl + (1 if rank < world - 1 else 0)
        dino_all = e.tower("dino", px_dino_local_halo)
        P = dino_all.shape[0] // n_d
        mns = cfg.get("max_num_segments", 24)
        if T <= mns + 1:
            seg_idx = list(range(T))
        else:
            if n_d >= 2:
                sims_local = e.sims_tensor(dino_all, n_d)
            else:
                sims_local = torch.zeros(0, dtype=torch.float32, device=dino_all.device)
            counts = [(h - l) - (0 if r < world - 1 else 1) for r, (l, h) in enumerate(ranges)]
            sims = self._all_gather_var(sims_local, counts).tolist()
            assert len(sims) == T - 1
            seg_idx = seg.select_segments(sims, mns)
        dino = dino_all[: Tl * P]
        # 2. local towers + connector
        sig = e.tower("siglip", px_siglip_local)
        X, sizes = e.connector(sig, dino, Tl, [tuple(image_size)] * Tl)
        N = X.shape[0] // Tl
        K = e.K
        # 3. shared plan, query hand-off
        max_visual_len = cfg["tokenizer_model_max_length"] - cfg.get("inference_max_length", 16) - n_text_tokens
        if cfg.get("query_type", "Avg_pool") != "Avg_pool":
            raise NotImplementedError("frame sharding implements the default query_type='Avg_pool' hand-off only")
        plan = seg.emit_plan(T, N, K, seg_idx, max_visual_len, cfg.get("add_static", True))
        pairs, comp_local = split_plan(plan, ranges, N, K)
        keys = plan["key_frames"]
        my_comp = [gi for gi, f in enumerate(plan["comp_frames"]) if lo <= f < hi]
        need = sorted(set(plan["comp_chunk"][gi] for gi in my_comp))
        owned = [c for c, s in enumerate(keys) if lo <= s < hi]
        q_owned = e.make_queries(X, N, N, [keys[c] - lo for c in owned]) if owned else None
        users = {}                               # chunk -> ranks that hold compressed frames of it
        for gi, f in enumerate(plan["comp_frames"]):
            users.setdefault(plan["comp_chunk"][gi], set()).add(owner_of(f, ranges))
        p2p, recv_buf = [], {}
        for ci, c in enumerate(owned):                       # ascending chunk id on both sides: pairwise order matches
            for r in sorted(users.get(c, ())):
                if r != rank:
                    p2p.append(dist.P2POp(dist.isend, q_owned[ci * K:(ci + 1) * K].contiguous(), r, self.group))
        for c in need:
            if not (lo <= keys[c] < hi):
                buf = torch.empty((K, e.query_width()), dtype=e.dtype, device=X.device)
                recv_buf[c] = buf
                p2p.append(dist.P2POp(dist.irecv, buf, owner_of(keys[c], ranges), self.group))
        if p2p:
            for rq in dist.batch_isend_irecv(p2p):
                rq.wait()
        comp = None
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
            comp = e.compress_frames(X, N, [plan["comp_frames"][gi] - lo for gi in my_comp], qtable, qsrc, prompt_ids)
        # 4. local emission + all-gather
        mine = pairs[rank]
        local = e.emit(X, comp, mine) if mine else torch.zeros(0, e.H, dtype=e.dtype, device=X.device)
        counts = [len(p) for p in pairs]
        return self._all_gather_var(local, counts)
