"""Tiled N-d ``Elemwise`` loop for gfx950: broadcasting / strided / transposed operands.

Replaces the general loop nest of ``Elemwise._c_all`` (pytensor/tensor/elemwise.py:848-1167;
``make_loop`` / ``make_reordered_loop`` in tensor/elemwise_cgen.py:212-465) for every launch that
is not the ``flat`` shape (all operands contiguous or scalar).  The reference reorders the loop
nest by stride and walks it element by element on one core; here the iteration space is cut into
``TR x TC`` tiles of (one outer dimension, the innermost output dimension):

* a workgroup owns one tile; the remaining outer dimensions ("batch") are decomposed ONCE per
  workgroup from ``blockIdx.x`` (wave-uniform arithmetic) — no division or modulo per element;
* ``TX`` lanes cover the tile's columns with ``V``-element packs (up to 16 bytes per lane and
  operand, coalesced along the contiguous axis), ``TY = 256 / TX`` thread rows times ``RPT`` rows
  per thread cover its rows; every load of a thread is issued before the first scalar graph runs;
* per-operand classes, decided on the host from the collapsed strides:

  ``V``  unit stride along the inner dimension: one pack load per row;
  ``R``  unit inner stride, zero row stride (a row vector broadcast down the rows): ONE pack load
         per thread, held in registers for all its rows;
  ``B``  zero inner stride (a column broadcast along the rows): one scalar load per row;
  ``T``  unit stride along the tile's ROW dimension and a large inner stride (a transposed
         operand): the tile is read along the operand's own contiguous axis (64 lanes = 512
         contiguous bytes), staged through LDS with an odd pitch and read back transposed;
  ``G``  anything else (strided, reversed): scalar loads at the operand's strides;
  ``S`` / ``C``  a single element on the device / a host-known scalar in the argument block.

Outputs are C-contiguous over the iteration space (pack stores), or reduced per workgroup
(``reduce_spec``: full reductions fused behind a broadcasting loop, as in ``flat``).
"""

from __future__ import annotations

from pytensor_amd.codegen import (
    BLOCK,
    CTYPE,
    PT_PAIR_HELPERS,
    REDUCE_OPS,
    VEC_HELPERS,
    _reduce_epilogue,
    _stream_load,
    _vec_type,
    emit_body,
    prelude_for,
    reduce_header,
)

MAX_BATCH = 3  # outer dimensions beyond the tile's row dimension (MAX_ND = 5 collapsed dims)


def tile_params(body, cls, nb, reduce_spec):
    P = ["long long R", "long long D", "long long nrb", "long long ncb"]
    P += [f"long long b{j}" for j in range(nb)]
    P += ["long long osr"] + [f"long long osb{j}" for j in range(nb)]
    for k, dt in enumerate(body["in_dtypes"]):
        if cls[k] == "C":
            P.append(f"const long long in{k}")
            continue
        P.append(f"const {CTYPE[dt]}* __restrict__ in{k}")
        P += [f"long long s{k}_b{j}" for j in range(nb)] + [f"long long s{k}_r", f"long long s{k}_i"]
    for k, dt in enumerate(body["out_dtypes"]):
        if reduce_spec[k] is None:
            P.append(f"{CTYPE[dt]}* __restrict__ out{k}")
        else:
            P.append(f"{CTYPE[reduce_spec[k][1]]}* __restrict__ part{k}")
    return P


def tile_kernel_source(name: str, body: dict, cls: str, nb: int, V: int, TX: int, RPT: int, reduce_spec=None, lds_rows: int = 0, tvec: int = 1) -> str:
    """``cls[k]`` ∈ 'VRBTGSC' per input (module docstring); ``nb`` batch dimensions; tile =
    ``TY*RPT`` rows x ``TX*V`` columns with ``TY = 256 // TX``.  ``lds_rows`` = the tile's row count
    when 'T' operands are present (then ``TY*RPT == lds_rows``)."""
    nin, nout = len(body["in_dtypes"]), len(body["out_dtypes"])
    reduce_spec = reduce_spec or [None] * nout
    TY = BLOCK // TX
    TC, TR = TX * V, TY * RPT
    assert TX * TY == BLOCK and len(cls) == nin
    if "T" in cls:
        assert lds_rows == TR
    P = tile_params(body, cls, nb, reduce_spec)
    L = [reduce_header() if any(reduce_spec) else "", prelude_for(body), VEC_HELPERS]
    L.append(f'extern "C" __global__ __launch_bounds__({BLOCK}) void {name}({", ".join(P)}) {{')
    L.append(f"  constexpr int TX = {TX}, TY = {TY}, RPT = {RPT}, V = {V}, TC = {TC}, TR = {TR};")
    # ---- workgroup-uniform decomposition of the tile index ----
    # (32-bit: the grid has fewer than 2^31 tiles; a 64-bit division is ~10x the instructions)
    L.append("  unsigned pt_t = blockIdx.x;")
    L.append("  const long long cb = pt_t % (unsigned)ncb; pt_t /= (unsigned)ncb;")
    L.append("  const long long rb = pt_t % (unsigned)nrb; pt_t /= (unsigned)nrb;")
    for j in range(nb - 1, 0, -1):
        L.append(f"  const long long q{j} = pt_t % (unsigned)b{j}; pt_t /= (unsigned)b{j};")
    if nb:
        L.append("  const long long q0 = pt_t;")
    L.append("  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;")
    L.append("  const long long col = cb * TC + (long long)tx * V;")
    L.append("  const bool cok = col < D;")
    L.append("  const long long colc = cok ? col : 0;")
    L.append("  const long long row0 = rb * TR;")
    ob = " + ".join(f"q{j} * osb{j}" for j in range(nb)) or "0"
    L.append(f"  const long long ob = {ob};")
    for k, dt in enumerate(body["in_dtypes"]):
        ct = CTYPE[dt]
        c = cls[k]
        if c == "C":
            L.append(f"  {ct} sc{k}; {{ const long long b = in{k}; __builtin_memcpy(&sc{k}, &b, sizeof({ct})); }}")
            continue
        if c == "S":
            L.append(f"  const {ct} sc{k} = in{k}[0];")
            continue
        off = " + ".join(f"q{j} * s{k}_b{j}" for j in range(nb)) or "0"
        L.append(f"  const {ct}* __restrict__ p{k} = in{k} + ({off});")
        if c == "R":
            if V > 1:
                L.append(f"  const {_vec_type(ct, V)} h{k} = *reinterpret_cast<const {_vec_type(ct, V)}*>(p{k} + colc);")
            else:
                L.append(f"  const {ct} h{k} = p{k}[colc];")
    for k, rs in enumerate(reduce_spec):
        if rs is not None:
            act = CTYPE[rs[1]]
            L.append(f"  {act} acc{k}_0 = pthip_dev::{REDUCE_OPS[rs[0]]}::identity<{act}>();")
    # ---- load phase: every row's operands requested before the first scalar graph ----
    L.append("  bool ok[RPT];")
    L.append("  long long rowv[RPT];")
    for k, dt in enumerate(body["in_dtypes"]):
        ct = CTYPE[dt]
        c = cls[k]
        if c == "V":
            L.append(f"  {_vec_type(ct, V) if V > 1 else ct} a{k}[RPT];")
        elif c == "B":
            L.append(f"  {ct} a{k}[RPT];")
        elif c == "G":
            L.append(f"  {ct} a{k}[RPT][V];")
    L.append("#pragma unroll\n  for (int i = 0; i < RPT; i++) {")
    L.append("    const long long row = row0 + ty + i * TY;")
    L.append("    ok[i] = cok && row < R;")
    L.append("    rowv[i] = row;")
    L.append("    const long long rowc = row < R ? row : R - 1;")
    for k, dt in enumerate(body["in_dtypes"]):
        ct = CTYPE[dt]
        c = cls[k]
        if c == "V":
            if V > 1:
                L.append(f"    a{k}[i] = {_stream_load(f'reinterpret_cast<const {_vec_type(ct, V)}*>(p{k} + rowc * s{k}_r + colc)', struct=True)};")
            else:
                L.append(f"    a{k}[i] = p{k}[rowc * s{k}_r + colc];")
        elif c == "B":
            L.append(f"    a{k}[i] = p{k}[rowc * s{k}_r];")
        elif c == "G":
            L.append(f"#pragma unroll\n    for (int e = 0; e < V; e++) a{k}[i][e] = p{k}[rowc * s{k}_r + (colc + e) * s{k}_i];")
    L.append("  }")
    # ---- transposed operands: the tile through LDS ----
    tks = [k for k in range(nin) if cls[k] == "T"]
    for k in tks:
        ct = CTYPE[body["in_dtypes"][k]]
        L.append(f"  __shared__ {ct} lds{k}[TC * (TR + 1)];")
    if tks:
        L.append("  {")
        L.append("    const long long cbase = cb * TC;")
        if tvec > 1:
            # ``tvec`` elements (16 bytes) per load along the operand's contiguous axis — the tile's ROW dimension (round 6:
            # 8-byte loads were half the bytes per instruction of every other stream of the kernel; A + B.T 4096^2 fp64
            # 57 -> 6x % of HBM cold).  The host guarantees R % tvec == 0, 16-byte aligned bases and strides.
            L.append(f"#pragma unroll\n    for (int l = threadIdx.x; l < (TR / {tvec}) * TC; l += {BLOCK}) {{")
            L.append(f"      const int rr = (l % (TR / {tvec})) * {tvec}, cc = l / (TR / {tvec});")
            L.append(f"      long long r_ = row0 + rr; r_ = r_ < R ? r_ : R - {tvec};")
            L.append("      long long c_ = cbase + cc; c_ = c_ < D ? c_ : D - 1;")
            for k in tks:
                ct = CTYPE[body["in_dtypes"][k]]
                L.append(f"      const {_vec_type(ct, tvec)} tv{k} = *reinterpret_cast<const {_vec_type(ct, tvec)}*>(p{k} + r_ * s{k}_r + c_ * s{k}_i);")
                L.append(f"#pragma unroll\n      for (int e = 0; e < {tvec}; e++) lds{k}[cc * (TR + 1) + rr + e] = tv{k}.v[e];")
            L.append("    }")
        else:
            L.append(f"#pragma unroll\n    for (int l = threadIdx.x; l < TR * TC; l += {BLOCK}) {{")
            L.append("      const int rr = l % TR, cc = l / TR;")
            L.append("      long long r_ = row0 + rr; r_ = r_ < R ? r_ : R - 1;")
            L.append("      long long c_ = cbase + cc; c_ = c_ < D ? c_ : D - 1;")
            for k in tks:
                L.append(f"      lds{k}[cc * (TR + 1) + rr] = p{k}[r_ * s{k}_r + c_ * s{k}_i];")
            L.append("    }")
        L.append("  }")
        L.append("  __syncthreads();")
    L.append("  __builtin_amdgcn_sched_barrier(0);")
    # ---- compute + store ----
    L.append("#pragma unroll\n  for (int i = 0; i < RPT; i++) {")
    for k, dt in enumerate(body["out_dtypes"]):
        if reduce_spec[k] is None and V > 1:
            L.append(f"    {_vec_type(CTYPE[dt], V)} r{k};")
    L.append("#pragma unroll\n    for (int e = 0; e < V; e++) {")
    in_names = []
    for k in range(nin):
        c = cls[k]
        if c in "CS":
            in_names.append(f"sc{k}")
        elif c == "V":
            in_names.append(f"a{k}[i].v[e]" if V > 1 else f"a{k}[i]")
        elif c == "R":
            in_names.append(f"h{k}.v[e]" if V > 1 else f"h{k}")
        elif c == "B":
            in_names.append(f"a{k}[i]")
        elif c == "G":
            in_names.append(f"a{k}[i][e]")
        elif c == "T":
            in_names.append(f"lds{k}[(tx * V + e) * (TR + 1) + ty + i * TY]")
        else:
            raise ValueError(c)
    out_names = []
    for k, dt in enumerate(body["out_dtypes"]):
        if reduce_spec[k] is None and V > 1:
            out_names.append(f"r{k}.v[e]")
        else:
            L.append(f"      {CTYPE[dt]} o{k};")
            out_names.append(f"o{k}")
    L.append(emit_body(body, in_names, out_names))
    for k, rs in enumerate(reduce_spec):
        if rs is not None:
            L.append(f"      if (ok[i]) acc{k}_0 = pthip_dev::{REDUCE_OPS[rs[0]]}::apply(acc{k}_0, ({CTYPE[rs[1]]})o{k});")
        elif V == 1:
            L.append(f"      if (ok[i]) out{k}[ob + rowv[i] * osr + col] = o{k};")
    L.append("    }")
    if V > 1:
        for k, dt in enumerate(body["out_dtypes"]):
            if reduce_spec[k] is None:
                L.append(f"    if (ok[i]) *reinterpret_cast<{_vec_type(CTYPE[dt], V)}*>(out{k} + ob + rowv[i] * osr + col) = r{k};")
    L.append("  }")
    L.append(_reduce_epilogue(reduce_spec, 1))
    L.append("}")
    return "\n".join(L)


# ---------------------------------------------------------------------------------------------
# N-d reductions over SOME dimensions (CAReduce with an axis tuple, optionally behind a fused
# scalar graph): the same tile, with the row and / or inner dimension reduced
# ---------------------------------------------------------------------------------------------

ND_REDUCE_OPS = {**REDUCE_OPS, "ScalarMaximum": "OpMax", "ScalarMinimum": "OpMin", "AND": "OpAnd", "OR": "OpOr", "XOR": "OpXor",
                 # log(sum(exp(.))) as ONE reduction over the running pair (max, scaled sum) — csrc/reduce_device.h pt_lse / OpLse
                 "LogSumExp": "OpLse"}


def _acc_ctype(op, acc):
    return f"pthip_dev::pt_lse<{CTYPE[acc]}>" if op == "LogSumExp" else CTYPE[acc]


def _lse_combine(L, k, T, sct, row_kept, inner_kept, TX, TY):
    """Fold the per-thread (max, scaled sum) pairs of a log-sum-exp output and store log(s) + m.  Never a butterfly of
    pair MERGES (two exps per lane and step — for 8192 x 2048 fp64 the combine cost more than the pass over the data):
    the maximum first (a plain butterfly), ONE rescale of each lane's sum to it, then a plain sum."""
    ex = "pt_lse_e" if T == "double" else "pthip_dev::lse_exp"
    lg = "pthip_dev::lse_log"
    mx, ad = f"pt_lse_max{k}", "pthip_dev::OpAdd"  # (plain max: a NaN state travels in the sums)
    L.append(f"  struct pt_lse_max{k} {{ static __device__ __forceinline__ {T} apply({T} a, {T} b) {{ return __builtin_fmax{'' if T == 'double' else 'f'}(a, b); }} }};")
    NW = BLOCK // 64

    def lanes(var_m, var_s, width, ind):  # fold (var_m, var_s) over `width` consecutive lanes (width <= 64, a power of two)
        L.append(f"{ind}{T} M_ = {var_m};")
        L.append(f"#pragma unroll\n{ind}for (int off = {width} / 2; off > 0; off >>= 1) M_ = {mx}::apply(M_, pthip_dev::shfl_xor_any(M_, off));")
        L.append(f"{ind}{T} S_ = {var_s} * {ex}(({var_m} == M_) ? ({T})0 : {var_m} - M_);")
        L.append(f"#pragma unroll\n{ind}for (int off = {width} / 2; off > 0; off >>= 1) S_ += pthip_dev::shfl_xor_any(S_, off);")

    if row_kept and not inner_kept:
        if TX == BLOCK:
            L.append(f"  __shared__ {T} smm{k}[{NW}], sms{k}[{NW}];")
        L.append("#pragma unroll\n  for (int i = 0; i < RPT; i++) {")
        lanes(f"acc{k}[i][0].m", f"acc{k}[i][0].s", min(TX, 64), "    ")
        if TX == BLOCK:
            L.append("    __syncthreads();")
            L.append(f"    if ((threadIdx.x & 63) == 0) {{ smm{k}[threadIdx.x >> 6] = M_; sms{k}[threadIdx.x >> 6] = S_; }}")
            L.append("    __syncthreads();")
            L.append(f"    M_ = smm{k}[0];")
            L.append(f"#pragma unroll\n    for (int q = 1; q < {NW}; q++) M_ = {mx}::apply(M_, smm{k}[q]);")
            L.append(f"    S_ = 0;")
            L.append(f"#pragma unroll\n    for (int q = 0; q < {NW}; q++) S_ += sms{k}[q] * {ex}((smm{k}[q] == M_) ? ({T})0 : smm{k}[q] - M_);")
        L.append("    const long long row = rb * TR + ty + i * TY;")
        L.append(f"    if (tx == 0 && row < R) dst{k}[ob + row * osr] = ({sct})({lg}(S_) + M_);")
        L.append("  }")
    elif inner_kept and not row_kept:
        if TY > 1:
            L.append(f"  __shared__ {T} smm{k}[TY][TC + 1], sms{k}[TY][TC + 1];")
            L.append(f"#pragma unroll\n  for (int e = 0; e < V; e++) {{ smm{k}[ty][tx * V + e] = acc{k}[0][e].m; sms{k}[ty][tx * V + e] = acc{k}[0][e].s; }}")
            L.append("  __syncthreads();")
        L.append("  if (ty == 0) {")
        L.append("    const long long col = cb * TC + (long long)tx * V;")
        L.append("#pragma unroll\n    for (int e = 0; e < V; e++) {")
        if TY > 1:
            L.append(f"      {T} M_ = smm{k}[0][tx * V + e];")
            L.append(f"      for (int y = 1; y < TY; y++) M_ = {mx}::apply(M_, smm{k}[y][tx * V + e]);")
            L.append(f"      {T} S_ = 0;")
            L.append(f"      for (int y = 0; y < TY; y++) S_ += sms{k}[y][tx * V + e] * {ex}((smm{k}[y][tx * V + e] == M_) ? ({T})0 : smm{k}[y][tx * V + e] - M_);")
        else:
            L.append(f"      const {T} M_ = acc{k}[0][e].m, S_ = acc{k}[0][e].s;")
        L.append(f"      if (col + e < D) dst{k}[ob + (col + e) * osi] = ({sct})({lg}(S_) + M_);")
        L.append("    }")
        L.append("  }")
    elif not row_kept and not inner_kept:
        L.append(f"  __shared__ {T} smm{k}[{NW}], sms{k}[{NW}];")
        L.append("  {")
        lanes(f"acc{k}[0][0].m", f"acc{k}[0][0].s", 64, "    ")
        L.append(f"    if ((threadIdx.x & 63) == 0) {{ smm{k}[threadIdx.x >> 6] = M_; sms{k}[threadIdx.x >> 6] = S_; }}")
        L.append("    __syncthreads();")
        L.append("    if (threadIdx.x == 0) {")
        L.append(f"      M_ = smm{k}[0];")
        L.append(f"#pragma unroll\n      for (int q = 1; q < {NW}; q++) M_ = {mx}::apply(M_, smm{k}[q]);")
        L.append("      S_ = 0;")
        L.append(f"#pragma unroll\n      for (int q = 0; q < {NW}; q++) S_ += sms{k}[q] * {ex}((smm{k}[q] == M_) ? ({T})0 : smm{k}[q] - M_);")
        L.append(f"      dst{k}[ob] = ({sct})({lg}(S_) + M_);")
        L.append("    }")
        L.append("  }")
    else:
        raise ValueError("nothing reduced in the tile")


def tile_reduce_params(body, cls, nkb, nrd, outs, finish=None):
    P = ["long long R", "long long D", "long long nrb", "long long ncb", "long long iters", "long long chunk", "long long ps_split"]
    P += [f"long long kb{j}" for j in range(nkb)] + [f"long long rd{j}" for j in range(nrd)]
    P += ["long long osr", "long long osi"] + [f"long long oskb{j}" for j in range(nkb)]
    for k, dt in enumerate(body["in_dtypes"]):
        if cls[k] == "C":
            P.append(f"const long long in{k}")
            continue
        P.append(f"const {CTYPE[dt]}* __restrict__ in{k}")
        P += [f"long long s{k}_kb{j}" for j in range(nkb)] + [f"long long s{k}_rd{j}" for j in range(nrd)] + [f"long long s{k}_r", f"long long s{k}_i"]
    for k, (op, acc, odt) in enumerate(outs):
        P.append(f"{CTYPE[odt]}* __restrict__ dst{k}")
    if finish:
        # one-pass finish (tile_reduce_source): the final outputs and their element strides, the per-group tickets,
        # the number of splits, lanes per output element of the closing fold, the device status word
        for k, fdt in enumerate(finish):
            P.append(f"{CTYPE[fdt]}* __restrict__ fin{k}")
        P += ["long long fosr", "long long fosi"] + [f"long long foskb{j}" for j in range(nkb)]
        P += ["int* __restrict__ pt_ticket", "long long nsplit", "long long n_nat", "int pt_L", "int* __restrict__ pt_status"]
    return P


def tile_reduce_source(name: str, body: dict, cls: str, nkb: int, nrd: int, row_kept: bool, inner_kept: bool, V: int, TX: int, RPT: int, outs, ui: int = 0, finish=None) -> str:
    """``out[kept] = reduce_{reduced} body(operands)`` — CAReduce over an axis tuple (pytensor/tensor/elemwise.py:1233,
    perform 1493-1511; the reference's C loop nest: elemwise.py:1520-1678, elemwise_cgen.py:467-761), any operand
    strides, with the scalar graph of a producing ``Elemwise`` evaluated on the fly.

    The iteration space is the tile of :func:`tile_kernel_source` — (row dimension, inner dimension) + batch
    dimensions — where each dimension is either kept or reduced:

    * kept batch dimensions (``nkb``), the row blocks of a kept row dimension and the column blocks of a kept
      inner dimension come from ``blockIdx.x``; a last factor of ``blockIdx.x`` is the SPLIT index;
    * everything reduced — reduced batch dimensions (``nrd``), the row blocks of a reduced row dimension, the
      column blocks of a reduced inner dimension — is one flat loop of ``iters`` tile visits, of which a
      workgroup takes ``chunk`` consecutive ones (split s: [s*chunk, (s+1)*chunk));
    * per-thread accumulators in the accumulator dtype (one per kept row of the thread x kept element of its
      pack), combined in a fixed order at the end: across the ``TX`` lanes of a row when the inner dimension
      is reduced (wave64 butterfly within ``TX <= 64`` lanes, LDS across waves for ``TX = 256``), across the
      ``TY`` thread rows through LDS when the row dimension is reduced.  No atomics: deterministic.

    ``outs[k] = (op, acc_dtype, store_dtype)``: with one split the store dtype is the output dtype and ``dst``
    the output; with several, ``dst`` is the partial array (``ps_split`` = the split stride; the output strides arrive pre-multiplied by the
    output's stride in it) in the accumulator dtype and a second launch (csrc/reduce.hip) folds the splits.
    """
    nin, nout = len(body["in_dtypes"]), len(body["out_dtypes"])
    assert len(outs) == nout and len(cls) == nin and "T" not in cls
    TY = BLOCK // TX
    TC, TR = TX * V, TY * RPT
    assert TX * TY == BLOCK
    if not inner_kept:
        assert TX <= 64 or TX == BLOCK
    NI = RPT if row_kept else 1  # accumulators per thread: kept rows x kept pack elements
    NE = V if inner_kept else 1
    P = tile_reduce_params(body, cls, nkb, nrd, outs, finish)
    if finish:
        assert not any(op == "LogSumExp" for op, _, _ in outs)
    L = [reduce_header(), prelude_for(body), VEC_HELPERS, PT_PAIR_HELPERS if finish else ""]
    L.append(f'extern "C" __global__ __launch_bounds__({BLOCK}) void {name}({", ".join(P)}) {{')
    L.append(f"  constexpr int TX = {TX}, TY = {TY}, RPT = {RPT}, V = {V}, TC = {TC}, TR = {TR};")
    L.append("  unsigned pt_t = blockIdx.x;")
    if inner_kept:
        L.append("  const long long cb = pt_t % (unsigned)ncb; pt_t /= (unsigned)ncb;")
    if row_kept:
        L.append("  const long long rb = pt_t % (unsigned)nrb; pt_t /= (unsigned)nrb;")
    for j in range(nkb - 1, -1, -1):
        L.append(f"  const long long q{j} = pt_t % (unsigned)kb{j}; pt_t /= (unsigned)kb{j};")
    L.append("  const long long split = pt_t;")
    L.append("  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;")
    for k, (op, acc, odt) in enumerate(outs):
        act = _acc_ctype(op, acc)
        L.append(f"  {act} acc{k}[{NI}][{NE}];")
        L.append(f"#pragma unroll\n  for (int i = 0; i < {NI}; i++)\n#pragma unroll\n    for (int e = 0; e < {NE}; e++) acc{k}[i][e] = pthip_dev::{ND_REDUCE_OPS[op]}::identity<{act}>();")
    for k, dt in enumerate(body["in_dtypes"]):
        ct = CTYPE[dt]
        if cls[k] == "C":
            L.append(f"  {ct} sc{k}; {{ const long long b = in{k}; __builtin_memcpy(&sc{k}, &b, sizeof({ct})); }}")
        elif cls[k] == "S":
            L.append(f"  const {ct} sc{k} = in{k}[0];")
        else:
            off = " + ".join(f"q{j} * s{k}_kb{j}" for j in range(nkb)) or "0"
            L.append(f"  const {ct}* __restrict__ pk{k} = in{k} + ({off});")
    if any(op == "LogSumExp" and acc == "float64" for op, acc, _ in outs):
        L.append("  const pt_expk pt_ek = pt_expk_load();  // the exp's constants, in registers for every instance below")
        L.append("  auto pt_lse_e = [&](double x) { return pt_exp_k(x, pt_ek); };")
    L.append("  long long it1 = (split + 1) * chunk; it1 = it1 < iters ? it1 : iters;")
    # UI tile visits per trip: the loads of all of them are issued before the first scalar graph runs (a visit of a
    # one-row tile is ONE pack per thread and operand; the chip needs ~64 KB in flight per CU)
    UI = int(ui) or max(1, 8 // RPT)
    L.append(f"  constexpr int UI = {UI};")
    L.append("  for (long long it = split * chunk; it < it1; it += UI) {")
    L.append("    bool ok[UI][RPT];")
    for k, dt in enumerate(body["in_dtypes"]):
        ct = CTYPE[dt]
        c = cls[k]
        if c == "V":
            L.append(f"    {_vec_type(ct, V) if V > 1 else ct} a{k}[UI][RPT];")
        elif c == "B":
            L.append(f"    {ct} a{k}[UI][RPT];")
        elif c == "G":
            L.append(f"    {ct} a{k}[UI][RPT][V];")
        elif c == "R":
            L.append(f"    {_vec_type(ct, V) if V > 1 else ct} h{k}[UI];")
    nvar = (0 if inner_kept else 1) + (0 if row_kept else 1) + nrd
    L.append("#pragma unroll\n    for (int w = 0; w < UI; w++) {")
    L.append("      const bool live = it + w < it1;")
    L.append("      unsigned u = (unsigned)(live ? it + w : it1 - 1);")
    left = nvar

    def take(var, ext):
        nonlocal left
        left -= 1
        if left == 0:
            L.append(f"      const long long {var} = u;")
        else:
            L.append(f"      const long long {var} = u % (unsigned){ext}; u /= (unsigned){ext};")

    if not inner_kept:
        take("cb", "ncb")
    if not row_kept:
        take("rb", "nrb")
    for j in range(nrd - 1, -1, -1):
        take(f"z{j}", f"rd{j}")
    L.append("      const long long col = cb * TC + (long long)tx * V;")
    L.append("      const bool cok = live && col < D;")
    L.append("      const long long colc = col < D ? col : 0;")
    L.append("      const long long row0 = rb * TR;")
    for k, dt in enumerate(body["in_dtypes"]):
        ct = CTYPE[dt]
        c = cls[k]
        if c in "CS":
            continue
        off = " + ".join(f"z{j} * s{k}_rd{j}" for j in range(nrd)) or "0"
        L.append(f"      const {ct}* __restrict__ p{k} = pk{k} + ({off});")
        if c == "R":
            if V > 1:
                L.append(f"      h{k}[w] = *reinterpret_cast<const {_vec_type(ct, V)}*>(p{k} + colc);")
            else:
                L.append(f"      h{k}[w] = p{k}[colc];")
    L.append("#pragma unroll\n      for (int i = 0; i < RPT; i++) {")
    L.append("        const long long row = row0 + ty + i * TY;")
    L.append("        ok[w][i] = cok && row < R;")
    L.append("        const long long rowc = row < R ? row : R - 1;")
    for k, dt in enumerate(body["in_dtypes"]):
        ct = CTYPE[dt]
        c = cls[k]
        if c == "V":
            if V > 1:
                L.append(f"        a{k}[w][i] = {_stream_load(f'reinterpret_cast<const {_vec_type(ct, V)}*>(p{k} + rowc * s{k}_r + colc)', struct=True)};")
            else:
                L.append(f"        a{k}[w][i] = p{k}[rowc * s{k}_r + colc];")
        elif c == "B":
            L.append(f"        a{k}[w][i] = p{k}[rowc * s{k}_r];")
        elif c == "G":
            L.append(f"#pragma unroll\n        for (int e = 0; e < V; e++) a{k}[w][i][e] = p{k}[rowc * s{k}_r + (colc + e) * s{k}_i];")
    L.append("      }")
    L.append("    }")
    # (the machine scheduler would otherwise sink every request next to its first use)
    L.append("    __builtin_amdgcn_sched_barrier(0);")
    # log-sum-exp outputs: the visit's values are kept (an element the tile does not cover counts as -inf: exp = 0) and
    # folded per accumulator in two steps — the local maximum, then INDEPENDENT exps against it (they pipeline), then one
    # merge into the running pair.  Pushing element by element is a chain of dependent exps per accumulator: measured 71 us
    # for 8192 x 2048 fp64 (profiles/r5q notes) against a 25 us read.
    lse = [k for k, (op, _, _) in enumerate(outs) if op == "LogSumExp"]
    for k in lse:
        L.append(f"    {CTYPE[outs[k][1]]} lv{k}[UI][RPT][V];")
    L.append("#pragma unroll\n    for (int w = 0; w < UI; w++) {")
    L.append("#pragma unroll\n    for (int i = 0; i < RPT; i++) {")
    L.append("#pragma unroll\n      for (int e = 0; e < V; e++) {")
    in_names = []
    for k in range(nin):
        c = cls[k]
        if c in "CS":
            in_names.append(f"sc{k}")
        elif c == "V":
            in_names.append(f"a{k}[w][i].v[e]" if V > 1 else f"a{k}[w][i]")
        elif c == "R":
            in_names.append(f"h{k}[w].v[e]" if V > 1 else f"h{k}[w]")
        elif c == "B":
            in_names.append(f"a{k}[w][i]")
        else:
            in_names.append(f"a{k}[w][i][e]")
    out_names = []
    for k, dt in enumerate(body["out_dtypes"]):
        L.append(f"        {CTYPE[dt]} o{k};")
        out_names.append(f"o{k}")
    L.append(emit_body(body, in_names, out_names, indent="        "))
    ia = "i" if row_kept else "0"
    ea = "e" if inner_kept else "0"
    for k, (op, acc, odt) in enumerate(outs):
        if op == "LogSumExp":
            act = CTYPE[acc]
            L.append(f"        lv{k}[w][i][e] = ok[w][i] ? ({act})o{k} : pthip_dev::Limits<{act}>::lowest();")
        else:
            L.append(f"        if (ok[w][i]) acc{k}[{ia}][{ea}] = pthip_dev::{ND_REDUCE_OPS[op]}::apply(acc{k}[{ia}][{ea}], ({CTYPE[acc]})o{k});")
    L.append("      }")
    L.append("    }")
    L.append("    }")
    for k in lse:
        act = CTYPE[outs[k][1]]
        ex = "pt_lse_e" if act == "double" else "pthip_dev::lse_exp"  # (the 24-instruction fp64 exp of the prelude, constants in registers)
        # members of accumulator (ia, ea): every (w, i, e) with i == ia when rows are kept, e == ea when the inner dim is
        L.append(f"#pragma unroll\n    for (int ia = 0; ia < {NI}; ia++)")
        L.append(f"#pragma unroll\n    for (int ea = 0; ea < {NE}; ea++) {{")
        L.append(f"      {act} ml = acc{k}[ia][ea].m;  // the new running maximum: the old one and every member")
        loop = ("#pragma unroll\n      for (int w = 0; w < UI; w++)\n#pragma unroll\n      for (int i = " + ("ia; i <= ia" if row_kept else "0; i < RPT") + "; i++)\n"
                "#pragma unroll\n      for (int e = " + ("ea; e <= ea" if inner_kept else "0; e < V") + "; e++)")
        # (a plain max: a NaN member is skipped here and reaches the result through exp(NaN - ml) in the sum — the
        #  NaN-propagating OpMax is three compares and six selects per element)
        L.append(loop + f" ml = __builtin_fmax{'' if act == 'double' else 'f'}(ml, lv{k}[w][i][e]);")
        # (x == ml -> exponent 0 -> exactly 1: covers equal infinities, whose difference is NaN, without a branch around the exp)
        L.append(f"      {act} sl = acc{k}[ia][ea].s * {ex}((acc{k}[ia][ea].m == ml) ? ({act})0 : acc{k}[ia][ea].m - ml);  // the old sum, rescaled")
        L.append(loop + f" sl += {ex}((lv{k}[w][i][e] == ml) ? ({act})0 : lv{k}[w][i][e] - ml);  // independent exps")
        L.append(f"      acc{k}[ia][ea] = pthip_dev::pt_lse<{act}>{{ml, sl}};")
        L.append("    }")
    L.append("  }")
    # ---- combine + store (fixed order: deterministic) ----
    obase = " + ".join(["split * ps_split"] + [f"q{j} * oskb{j}" for j in range(nkb)])
    L.append(f"  const long long ob = {obase};")
    for k, (op, acc, odt) in enumerate(outs):
        act, opn, sct = _acc_ctype(op, acc), f"pthip_dev::{ND_REDUCE_OPS[op]}", CTYPE[odt]
        if op == "LogSumExp":
            _lse_combine(L, k, CTYPE[acc], sct, row_kept, inner_kept, TX, TY)
            continue
        if row_kept and not inner_kept:
            if TX == BLOCK:
                L.append(f"  __shared__ {act} sm{k}[{BLOCK // 64}];")
            L.append(f"#pragma unroll\n  for (int i = 0; i < RPT; i++) {{")
            L.append(f"    {act} v = acc{k}[i][0];")
            if TX == BLOCK:
                L.append(f"    v = pthip_dev::block_reduce<{opn}, {act}, {BLOCK}>(v, sm{k});")
            else:
                L.append(f"#pragma unroll\n    for (int off = TX / 2; off > 0; off >>= 1) v = {opn}::apply(v, pthip_dev::shfl_xor_any(v, off));")
            L.append("    const long long row = rb * TR + ty + i * TY;")
            if finish:
                L.append(f"    if (tx == 0 && row < R) {{ pt_u64 b_ = 0; __builtin_memcpy(&b_, &v, sizeof(v)); pt_pair_store((pt_u64*)dst{k} + 2 * (ob + row * osr), b_, b_ ^ PT_PAIR_MAGIC); }}")
            else:
                L.append(f"    if (tx == 0 && row < R) dst{k}[ob + row * osr] = ({sct})v;")
            L.append("  }")
        elif inner_kept and not row_kept:
            if TY > 1:
                L.append(f"  __shared__ {act} sm{k}[TY][TC + 1];")
                L.append(f"#pragma unroll\n  for (int e = 0; e < V; e++) sm{k}[ty][tx * V + e] = acc{k}[0][e];")
                L.append("  __syncthreads();")
                L.append("  if (ty == 0) {")
                L.append(f"#pragma unroll\n    for (int e = 0; e < V; e++) {{")
                L.append(f"      {act} v = sm{k}[0][tx * V + e];")
                L.append(f"      for (int y = 1; y < TY; y++) v = {opn}::apply(v, sm{k}[y][tx * V + e]);")
                L.append(f"      acc{k}[0][e] = v;")
                L.append("    }")
                L.append("  }")
            L.append("  {")
            L.append("    const long long col = cb * TC + (long long)tx * V;")
            if finish:
                L.append(f"#pragma unroll\n    for (int e = 0; e < V; e++) if (ty == 0 && col + e < D) {{ pt_u64 b_ = 0; __builtin_memcpy(&b_, &acc{k}[0][e], sizeof(acc{k}[0][e])); pt_pair_store((pt_u64*)dst{k} + 2 * (ob + (col + e) * osi), b_, b_ ^ PT_PAIR_MAGIC); }}")
            else:
                L.append(f"#pragma unroll\n    for (int e = 0; e < V; e++) if (ty == 0 && col + e < D) dst{k}[ob + (col + e) * osi] = ({sct})acc{k}[0][e];")
            L.append("  }")
        elif not row_kept and not inner_kept:
            L.append(f"  __shared__ {act} sm{k}[{BLOCK // 64}];")
            L.append(f"  {{ const {act} v = pthip_dev::block_reduce<{opn}, {act}, {BLOCK}>(acc{k}[0][0], sm{k});")
            if finish:
                L.append(f"    if (threadIdx.x == 0) {{ pt_u64 b_ = 0; __builtin_memcpy(&b_, &v, sizeof(v)); pt_pair_store((pt_u64*)dst{k} + 2 * ob, b_, b_ ^ PT_PAIR_MAGIC); }} }}")
            else:
                L.append(f"    if (threadIdx.x == 0) dst{k}[ob] = ({sct})v; }}")
        else:
            raise ValueError("nothing reduced in the tile: use tile_kernel_source")
    if finish:
        # ---- ONE pass (round 6): every workgroup has published its partials as self-validating pairs (no fence: see
        # PT_PAIR_HELPERS); it takes a ticket of its GROUP — the workgroups that differ only in the split index — and the
        # last of the nsplit to arrive folds the group's partials into the final outputs: pt_L lanes per output element,
        # each lane the splits lane, lane + pt_L, ... in order, then a butterfly over the pt_L lanes (a fixed order
        # whichever workgroup is last), and zeroes the pairs for the next launch.  The second-stage launch (4.5 us + a
        # launch gap behind a 22 us pass over 134 MB: profiles/r8_hotpath_cold_kernel_stats.md) goes away.
        NEL = TR if (row_kept and not inner_kept) else (TC if (inner_kept and not row_kept) else 1)
        L.append("  __shared__ int pt_last_;")
        L.append("  __syncthreads();")
        L.append("  if (threadIdx.x == 0) pt_last_ = atomicAdd(pt_ticket + (blockIdx.x % (unsigned)n_nat), 1) == (int)nsplit - 1;")
        L.append("  __syncthreads();")
        L.append("  if (pt_last_) {")
        L.append("    if (threadIdx.x == 0) __hip_atomic_store(pt_ticket + (blockIdx.x % (unsigned)n_nat), 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);")
        L.append("    const long long ob0 = ob - split * ps_split;")
        fob = " + ".join(f"q{j} * foskb{j}" for j in range(nkb)) or "0"
        L.append(f"    const long long fob = {fob};")
        L.append("    const int ls_ = threadIdx.x % pt_L, es_ = threadIdx.x / pt_L, epp_ = 256 / pt_L;")
        L.append(f"    for (int e0_ = 0; e0_ < {NEL}; e0_ += epp_) {{")
        L.append("      const int el_ = e0_ + es_;")
        if row_kept and not inner_kept:
            L.append("      const long long pos_ = rb * TR + el_;")
            L.append(f"      const bool val_ = el_ < {NEL} && pos_ < R;")
            L.append("      const long long pidx_ = ob0 + pos_ * osr, fidx_ = fob + pos_ * fosr;")
        elif inner_kept and not row_kept:
            L.append("      const long long pos_ = cb * TC + el_;")
            L.append(f"      const bool val_ = el_ < {NEL} && pos_ < D;")
            L.append("      const long long pidx_ = ob0 + pos_ * osi, fidx_ = fob + pos_ * fosi;")
        else:
            L.append("      const bool val_ = el_ < 1;")
            L.append("      const long long pidx_ = ob0, fidx_ = fob;")
        for k, (op, acc, odt) in enumerate(outs):
            act, opn = _acc_ctype(op, acc), f"pthip_dev::{ND_REDUCE_OPS[op]}"
            L.append("      {")
            L.append(f"        {act} r_ = {opn}::identity<{act}>();")
            L.append("        pt_u64 b_[8]; bool got_[8];")
            L.append("#pragma unroll\n        for (int q = 0; q < 8; q++) { b_[q] = 0; got_[q] = !(val_ && ls_ + pt_L * q < nsplit); }")
            L.append("        for (long long spins = 0;; spins++) {")
            L.append("          bool all_ = true;")
            L.append("#pragma unroll\n          for (int q = 0; q < 8; q++)")
            L.append(f"            if (!got_[q]) {{ got_[q] = pt_pair_poll((const pt_u64*)dst{k} + 2 * (pidx_ + (long long)(ls_ + pt_L * q) * ps_split), b_[q]); all_ = all_ && got_[q]; }}")
            L.append("          if (all_) break;")
            L.append("          if (spins > (1ll << 22)) { atomicOr(pt_status, 16); break; }")
            L.append("        }")
            L.append("#pragma unroll\n        for (int q = 0; q < 8; q++)")
            L.append("          if (val_ && ls_ + pt_L * q < nsplit) {")
            L.append(f"            {act} v_; __builtin_memcpy(&v_, &b_[q], sizeof(v_));")
            L.append(f"            r_ = {opn}::apply(r_, v_);")
            L.append(f"            pt_pair_store((pt_u64*)dst{k} + 2 * (pidx_ + (long long)(ls_ + pt_L * q) * ps_split), 0, 0);  // clean for the next launch")
            L.append("          }")
            L.append(f"#pragma unroll\n        for (int off = 32; off > 0; off >>= 1) {{ const {act} o_ = pthip_dev::shfl_xor_any(r_, off); if (off < pt_L) r_ = {opn}::apply(r_, o_); }}")
            L.append(f"        if (ls_ == 0 && val_) fin{k}[fidx_] = ({CTYPE[finish[k]]})r_;")
            L.append("      }")
        L.append("    }")
        L.append("  }")
    L.append("}")
    return "\n".join(L)
