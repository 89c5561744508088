"""Host-side integer logic of the path (no tensors of size > T): frame budget and uniform sub-sampling, segment
selection from the adjacent-frame similarities, 8-frame chunk table, unpad geometry / SVA masks, and the index map
that lays the emitted tokens out.  Mirrors tdc/cambrian_arch.py (line numbers in each docstring); bit-exact by
construction (same integer arithmetic), similarity ranking uses a stable sort (ties -> lowest index)."""
import math


def get_max_num_frames(text_len, cfg):
    """tdc/cambrian_arch.py:748-780 (text_len = position of the first pad id, else len(input_ids))."""
    K = cfg.get("context_token_num", 16)
    if not cfg.get("audio_input", False):
        tpf = (144 + K * 7) // 8
    else:
        tpf = (144 + 50 + K * 7) // 8
    if not cfg.get("add_static", True):
        tpf = 16
    return max(1, (cfg["tokenizer_model_max_length"] - text_len - cfg.get("inference_max_length", 16)) // tpf)


def uniform_indices(T, max_frames):
    """tdc/cambrian_arch.py:910-912 / :815-817: int(T/max * i)."""
    if T <= max_frames:
        return list(range(T))
    interval = T / float(max_frames)
    return [int(interval * i) for i in range(max_frames)]


def select_segments(sims, max_num_segments=24):
    """argsort(sims)[:n].sort() (tdc/cambrian_arch.py:849) on a host list of floats; stable for ties."""
    order = sorted(range(len(sims)), key=lambda i: (sims[i], i))[:max_num_segments]
    return sorted(order)


def selection_band(sims, max_num_segments, eps):
    """Which adjacent-frame similarities must be known more precisely before the a5 selection (tdc/cambrian_arch.py:849) can be
    trusted, when `sims` are within `eps` of the values a more precise tower would give (bf16 DINOv2 operands against the
    reference's fp16: measured 3.6e-4 ... 4.2e-4).
    With v_n / v_n1 the n-th and (n+1)-th smallest value (n = max_num_segments): a pair with sims[i] < v_n1 - 2 eps is selected
    whatever its precise value is (every pair that could precede it precisely has sims < v_n1, and there are at most n of those,
    itself included); a pair with sims[i] > v_n + 2 eps never is (the n pairs with sims <= v_n are all precisely smaller).  What
    is left, v_n1 - 2 eps <= sims[i] <= v_n + 2 eps, is returned (ascending) and has to be re-ranked (`select_refined`); when
    v_n1 - v_n > 4 eps the interval is empty: no error of that size can swap a selected pair with an unselected one."""
    n = len(sims)
    if n <= max_num_segments or eps is None or eps <= 0:
        return []
    order = sorted(range(n), key=lambda i: (sims[i], i))
    v_n, v_n1 = sims[order[max_num_segments - 1]], sims[order[max_num_segments]]
    lo, hi = v_n1 - 2.0 * eps, v_n + 2.0 * eps
    if lo > hi:
        return []
    return [i for i in range(n) if lo <= sims[i] <= hi]


def select_refined(sims, max_num_segments, eps, band, refined):
    """The selection after the band's similarities were recomputed precisely (`refined[j]` belongs to pair `band[j]`): every pair
    below the band, plus the lowest of the band by (refined value, index) - the stable ranking of select_segments - up to
    max_num_segments.  With refined == the precise tower's values this is exactly what ranking ALL of that tower's similarities
    selects, as long as |sims - precise| <= eps (the precise top-n is a prefix of the precise order, so what it takes from the
    band is a prefix of the band's order)."""
    if not band:
        return select_segments(sims, max_num_segments)
    assert len(band) == len(refined)
    inband = set(band)
    order = sorted(range(len(sims)), key=lambda i: (sims[i], i))
    v_n1 = sims[order[max_num_segments]]
    low = [i for i in range(len(sims)) if sims[i] < v_n1 - 2.0 * eps and i not in inband]
    need = max_num_segments - len(low)
    assert 0 <= need <= len(band), (need, len(band), len(low))
    ranked = sorted(range(len(band)), key=lambda j: (refined[j], band[j]))[:need]
    return sorted(low + [band[j] for j in ranked])


def band_allowed(band, T, max_fraction):
    """Cost control of the refinement: the band's frames are re-encoded by a second tower, so a band that covers a large part of the
    video - a plateau of near-identical similarities at the decisive rank, e.g. a clip with fewer real scene changes than
    max_num_segments - would cost a large part of a tower pass to settle an order the reference itself does not define there (it ranks
    fp16 similarity VALUES, spaced 4.9e-4 in [0.5, 1), with an argsort that promises nothing among equal ones:
    tdc/cambrian_arch.py:841,849).  The band is refined when its frames number at most max(8, max_fraction * T)."""
    return bool(band) and len(band_frames(band)) <= max(8, int(max_fraction * T))


def band_frames(band):
    """frames whose precise features the band's pairs need: pair i = frames (i, i + 1); ascending, no duplicates"""
    return sorted({f for i in band for f in (i, i + 1)})


def chunk_table(T, seg_indices):
    """tdc/cambrian_arch.py:1541-1545, :1603-1608: segments split at seg+1, then <=8-frame chunks -> [(start, end)]."""
    pts = [0] + [int(s) + 1 for s in seg_indices] + [T]
    chunks = []
    for a, b in zip(pts[:-1], pts[1:]):
        for s in range(a, b, 8):
            chunks.append((s, min(s + 8, b)))
    return chunks


def unpad_bounds(cur_h, cur_w, image_size):
    """unpad_image (tdc/cambrian_arch.py:512-544): image_size is unpacked as (width, height) although callers pass
    (height, width) - reproduced.  Returns (r0, r1, c0, c1)."""
    ow, oh = image_size
    if ow / oh > cur_w / cur_h:
        new_h = int(oh * (cur_w / ow))
        pad = (cur_h - new_h) // 2
        return pad, cur_h - pad, 0, cur_w
    new_w = int(ow * (cur_h / oh))
    pad = (cur_w - new_w) // 2
    return 0, cur_h, pad, cur_w - pad


def window_mask_bytes(side, r, image_size):
    """kv mask of one frame for one tower as a [side*side][r*r] list of 0/1 (tdc/cambrian_arch.py:487-509,
    :619-669): padding rows/cols masked, all-masked windows forced to all-ones."""
    n = side * r
    ow, oh = image_size
    rows_ok = [1] * n
    cols_ok = [1] * n
    if ow / oh > 1.0:
        new_h = int(oh * (n / ow))
        pad = (n - new_h) // 2
        for i in range(pad):
            rows_ok[i] = 0
            rows_ok[n - 1 - i] = 0
    else:
        new_w = int(ow * (n / oh))
        pad = (n - new_w) // 2
        for i in range(pad):
            cols_ok[i] = 0
            cols_ok[n - 1 - i] = 0
    out = []
    for i in range(side):
        for j in range(side):
            m = [rows_ok[i * r + a] & cols_ok[j * r + b] for a in range(r) for b in range(r)]
            if sum(m) == 0:
                m = [1] * (r * r)
            out.append(m)
    return out


def unpad_newline_map(side, image_size, frame, newline_table=1, newline_row=0, feat_table=0):
    """gather map (table,row) pairs for one frame of tdc/cambrian_arch.py:1195-1293: side x side tokens, unpadded,
    one newline token appended to every kept row.  feat rows are frame*side*side + y*side + x."""
    r0, r1, c0, c1 = unpad_bounds(side, side, image_size)
    src = []
    for y in range(r0, r1):
        for x in range(c0, c1):
            src.append((feat_table, frame * side * side + y * side + x))
        src.append((newline_table, newline_row))
    return src, (r1 - r0, c1 - c0)


class Plan(dict):
    """emit_plan's result: a dict (chunks, comp_frames, comp_chunk, key_frames) plus the token stream as three numpy
    int arrays of equal length - kind (0 frame token, 1 context token, 2 separator), a (frame / compressed-frame index), b
    (token index) - and, on first access only, plan["src"]: the same stream as a list of ('f', frame, tok) / ('c',
    comp_idx, k) / ('s',) tuples (tests and small cases; a 512-frame video has ~75 k entries)."""

    def __missing__(self, key):
        if key != "src":
            raise KeyError(key)
        v = [("f", a, b) if k == 0 else ("c", a, b) if k == 1 else ("s",)
             for k, a, b in zip(self.kind.tolist(), self.a.tolist(), self.b.tolist())]
        self["src"] = v
        return v

    def __len__(self):          # number of ordinary keys; the stream length is len(plan.kind)
        return dict.__len__(self)


def emit_plan(T, N, K, seg_indices, max_visual_len, add_static=True):
    """Token layout of tdc/cambrian_arch.py:1603-1709 (add_sep is hard-wired True at :1512): returns a Plan with
       chunks          [(start,end)]
       comp_frames     frame index of every compressed frame, in emission order
       comp_chunk      for each compressed frame the index (among chunks that run the Q-Former) of its chunk
       key_frames      key frame index of every chunk that runs the Q-Former
       .kind/.a/.b     the emitted stream (see Plan) after the per-chunk tail clipping and the final [:max_visual_len] cut
    add_static=False (:1625-1628, :1686-1690): no static tokens; every frame of every chunk, the key frame and
    single-frame chunks included, is compressed."""
    import numpy as np
    chunks = chunk_table(T, seg_indices)
    comp_frames, comp_chunk, key_frames = [], [], []
    ar_n, ar_k = np.arange(N, dtype=np.int64), np.arange(K, dtype=np.int64)
    z1 = np.zeros(1, dtype=np.int64)
    kinds, aa, bb = [], [], []          # one array triple per chunk
    for (s, e) in chunks:
        k_parts, a_parts, b_parts = [], [], []
        if add_static:
            k_parts += [np.zeros(N, dtype=np.int64), z1 + 2]
            a_parts += [np.full(N, s, dtype=np.int64), z1]
            b_parts += [ar_n, z1]
        if e - s > 1 or not add_static:
            ci = len(key_frames)
            key_frames.append(s)
            f0 = s + 1 if add_static else s
            nf = e - f0
            if nf > 0:
                i0 = len(comp_frames)
                comp_frames.extend(range(f0, e))
                comp_chunk.extend([ci] * nf)
                blk_k = np.concatenate([np.ones(K, dtype=np.int64), z1 + 2])
                blk_b = np.concatenate([ar_k, z1])
                k_parts.append(np.tile(blk_k, nf))
                b_parts.append(np.tile(blk_b, nf))
                a_parts.append(np.repeat(np.arange(i0, i0 + nf, dtype=np.int64), K + 1))
        kinds.append(np.concatenate(k_parts) if k_parts else np.zeros(0, dtype=np.int64))
        aa.append(np.concatenate(a_parts) if a_parts else np.zeros(0, dtype=np.int64))
        bb.append(np.concatenate(b_parts) if b_parts else np.zeros(0, dtype=np.int64))
    total = sum(len(k) for k in kinds)
    if total > max_visual_len:
        rm = math.ceil((total - max_visual_len) / len(kinds))
        # python's t[:-rm]: an empty result when rm >= len(t)
        kinds = [k[:-rm] for k in kinds]
        aa = [a[:-rm] for a in aa]
        bb = [b[:-rm] for b in bb]
    # python slice semantics as in the reference (cambrian_arch.py:1709, new_visual_emb_frames[:max_visual_len]): a NEGATIVE
    # budget (text longer than the model length) drops the last |budget| tokens, it does not empty the stream
    cat = lambda xs: (np.concatenate(xs) if xs else np.zeros(0, dtype=np.int64))[:max_visual_len]
    plan = Plan(chunks=chunks, comp_frames=comp_frames, comp_chunk=comp_chunk, key_frames=key_frames)
    plan.kind, plan.a, plan.b = cat(kinds), cat(aa), cat(bb)
    plan.a = np.where(plan.kind == 2, 0, plan.a)
    plan.b = np.where(plan.kind == 2, 0, plan.b)
    return plan


def emit_pairs(plan, Nf, K):
    """(table, row) gather pairs of a Plan as an int32 numpy array [n, 2]: table 0 = frame tokens (row = frame * Nf + tok),
    1 = context tokens (row = comp_idx * K + k), 2 = the frame separator (row 0)."""
    import numpy as np
    out = np.zeros((len(plan.kind), 2), dtype=np.int32)
    out[:, 0] = plan.kind
    out[:, 1] = np.where(plan.kind == 0, plan.a * Nf + plan.b, np.where(plan.kind == 1, plan.a * K + plan.b, 0))
    return out


def audio_plan(window_sizes, sample_indices, dist=10):
    """a20 host logic (tdc/cambrian_arch.py:1552-1589): BEATs windows of `window_sizes[w]` tokens (50 per second) ->
    one entry per emitted audio frame.  entry = (parts, direct): parts = [(window, start, end), ...] token slices (a
    slice shorter than 50 is pooled to 50 first), direct = True when the reference appends the token verbatim,
    False when it average-pools the concatenation of the parts back to 50 tokens (kept second + dropped seconds)."""
    entries, seg = [], []
    for w, n_w in enumerate(window_sizes):
        k = w * dist
        window = sample_indices[k:k + dist]
        sample_len = len(window)
        for idx, ind in enumerate(window):
            s, e = idx * 50, min((idx + 1) * 50, n_w)
            if e - s <= 0:
                continue
            part = (w, s, e)
            if ind == 1:
                if seg:
                    entries.append((seg, False))
                    seg = []
                seg.append(part)
                if idx + 1 < sample_len and sample_indices[k + idx + 1] == 1:
                    entries.append(([part], True))
                    seg = []
            elif ind == 0:
                seg.append(part)
    if seg:
        entries.append((seg, False))
    return entries


def shard_ranges(T, world):
    """contiguous frame ranges per rank: rank r owns [lo, hi)."""
    base, rem = divmod(T, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out
