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


def tile_kernel_source(name: str, body: dict, cls: str, nb: int, V: int, TX: int, RPT: int, reduce_spec=None, lds_rows: int = 0) -> str:
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
        L.append(f"#pragma unroll\n    for (int l = threadIdx.x; l < TR * TC; l += {BLOCK}) {{")
        L.append("      const int rr = l % TR, cc = l / TR;")
        L.append("      long long r_ = row0 + rr; r_ = r_ < R ? r_ : R - 1;")
        L.append("      long long c_ = cbase + cc; c_ = c_ < D ? c_ : D - 1;")
        for k in tks:
            L.append(f"      lds{k}[cc * (TR + 1) + rr] = p{k}[r_ * s{k}_r + c_ * s{k}_i];")
        L.append("    }")
        L.append("  }")
        L.append("  __syncthreads();")
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
