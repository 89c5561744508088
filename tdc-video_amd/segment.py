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


def emit_plan(T, N, K, seg_indices, max_visual_len, add_static=True):
    """Token layout of tdc/cambrian_arch.py:1603-1709 (add_sep is hard-wired True at :1512): returns
       chunks          [(start,end)]
       comp_frames     frame index of every compressed frame, in emission order
       comp_chunk      for each compressed frame the index (among chunks that run the Q-Former) of its chunk
       key_frames      key frame index of every chunk that runs the Q-Former
       src             list of (kind, a, b): ('f', frame, tok) static token, ('c', comp_idx, k) context token, ('s',)
                        frame separator - after the per-chunk tail clipping and the final [:max_visual_len] cut.
    add_static=False (:1625-1628, :1686-1690): no static tokens; every frame of every chunk, the key frame and
    single-frame chunks included, is compressed."""
    chunks = chunk_table(T, seg_indices)
    comp_frames, comp_chunk, key_frames = [], [], []
    per_chunk = []
    for (s, e) in chunks:
        toks = [("f", s, t) for t in range(N)] + [("s",)] if add_static else []
        if e - s > 1 or not add_static:
            ci = len(key_frames)
            key_frames.append(s)
            for f in range(s + 1 if add_static else s, e):
                idx = len(comp_frames)
                comp_frames.append(f)
                comp_chunk.append(ci)
                toks += [("c", idx, k) for k in range(K)] + [("s",)]
        per_chunk.append(toks)
    total = sum(len(t) for t in per_chunk)
    if total > max_visual_len:
        rm = math.ceil((total - max_visual_len) / len(per_chunk))
        per_chunk = [t[:-rm] for t in per_chunk]
    src = [x for t in per_chunk for x in t][:max_visual_len]
    return dict(chunks=chunks, comp_frames=comp_frames, comp_chunk=comp_chunk, key_frames=key_frames, src=src)


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
