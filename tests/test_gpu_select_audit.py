"""The GEMM dispatcher's thresholds, made falsifiable (review of round 2, next #7).

`gemm.cpp::select` holds ~25 measured crossovers between ten kernels.  They were measured on the builder's boxes; on a box
where one of them is wrong, the symptom used to be a slower bench entry and nothing else.  This test times AUTO against EVERY
kernel that accepts the descriptor over a fixed grid of 64 bf16 shapes -- the bench's skinny / output-bound / mid-size / decode
shapes among them -- interleaved, three rounds, medians, cold operands (launches rotate through operand sets larger than the
Infinity Cache), and fails when AUTO is more than 10 % AND more than 2 us behind the best forced kernel on any shape (round 6, review of
round 5 next #8: the bar was 15 % + 3 us, wide enough to hide a wrong kernel on every launch under 20 us; 2 us: launches of 8-15 us carry
about +-1 us of launch-to-launch noise per kernel, and a suspect is re-measured over seven longer rounds before it counts).  The full
table goes to gpurun_out/select_audit.txt."""
import os
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]

GRID = [
    # the bench's entries
    (8192, 8192, 64), (64, 8192, 8192), (8192, 64, 8192), (1, 8192, 8192), (16, 8192, 8192), (16, 28672, 8192), (64, 28672, 8192),
    (128, 28672, 8192), (4096, 4096, 4096), (6144, 6144, 6144), (4608, 4096, 8192), (2048, 2048, 2048), (4096, 2048, 4096),
    # few rows / columns
    (2, 8192, 8192), (4, 8192, 8192), (4, 2048, 4096), (32, 14336, 4096), (48, 4096, 4096), (96, 8192, 4096), (8192, 128, 8192),
    (256, 12288, 4096), (128, 14336, 4096), (3072, 4, 8192),
    # short K over many tiles
    (8192, 8192, 256), (16384, 8192, 128), (8192, 3072, 512), (7168, 4096, 1024), (4096, 4096, 64),
    # mid-size: the 128x128 / 256x128 / 256x256 crossovers
    (1024, 4096, 4096), (2048, 2048, 8192), (2560, 2560, 4096), (4096, 2048, 2048), (3072, 3072, 3072), (4096, 1536, 8192),
    (4096, 2304, 4096), (3072, 2560, 1024),
    # partly filled rounds of the 256x256 tile
    (5120, 5120, 5120), (4352, 4096, 4096), (8192, 8192, 2048), (6144, 4096, 2048),
    # round 5: the cost table's band (one round of 256 x 256 / 256 x 192 / 192 x 192 / 256 x 128 / 128 x 128 tiles) and the 16x16x32 kernel's ground
    (2304, 2304, 2304), (2560, 2560, 1024), (4096, 2048, 1024), (3072, 3072, 1024), (2048, 3072, 8192), (4096, 3072, 4096), (3584, 3584, 3584),
    (1792, 4864, 4096), (2048, 4608, 1024), (2816, 2816, 4096), (1920, 1920, 4096), (8192, 8192, 8192),
    # late round 5: the tables' third round and K = 512 over several rounds, the 65-128-row band, one or two K-tiles along a long side
    (4672, 7360, 3072), (6144, 6144, 1024), (3520, 10112, 512), (128, 16384, 512), (16384, 104, 1024), (116, 40960, 2048), (29512, 32, 128),
    (5, 53432, 1024),
    # late round 6: the rules eight fresh audit seeds produced (profiles/dispatch_rules.md STREAM_PART_TILES_*, STREAM_PART_ROUND_*, STREAM_COLS_K_MAX,
    # NARROW_TILE_MIN_KTILES), each at a shape its A/B sweep decided by 15 % or more
    (18, 37312, 2048), (47, 8704, 3072), (19152, 6, 14336), (8840, 960, 512),
]
ALGOS = ["auto", "lp128", "lp256x128", "lp256w4", "lp256p", "lp256q", "stream64", "skinny", "lp256x192", "lp192x192", "lp256m16", "lp256qm"]
# few rows against a ROW-MAJOR [K][N] weight (review of round 3, next #6): the rhs layout TensorHandle::new_contiguous gives
GRID_NN = [(1, 8192, 8192), (4, 8192, 8192), (8, 8192, 8192), (16, 8192, 8192), (4, 4096, 4096), (4, 14336, 4096), (4, 4096, 14336),
           (16, 4096, 14336), (16, 28672, 8192), (4, 32000, 4096), (1, 128256, 4096), (16, 128256, 4096), (32, 8192, 8192), (16, 14336, 4096),
           (16, 16384, 4096), (12, 8192, 8192), (16, 6144, 6144), (16, 12288, 4096)]
ALGOS_NN = ["auto", "lp128", "nnrows"]
# f32 operands with few rows or columns (round 5: the streaming kernel's f32 form between the FMA kernel and the 128 x 128 f32 tile)
GRID_F32 = [(1, 8192, 8192), (4, 8192, 8192), (8, 8192, 8192), (16, 8192, 8192), (32, 4096, 4096), (64, 8192, 8192), (16, 28672, 4096),
            (4096, 32, 4096), (16, 1024, 1024), (8192, 8, 8192), (128, 4096, 4096)]
ALGOS_F32 = ["auto", "f32", "skinny", "stream64"]
# late round 6: the layouts / types the seeded audits had never run -- each shape decided by 15 % or more in its audit (profiles/dispatch_rules.md)
GRID_F32_TILES = [(4672, 3968, 4096), (4160, 4096, 4096), (3072, 3072, 3072), (96, 57504, 512), (27, 18104, 64), (7, 49648, 64), (15, 14168, 64)]   # f32 tile table, short-K rules
ALGOS_F32_TILES = ["auto", "f32", "lp256w4", "stream64"]
GRID_C32 = [(7680, 3776, 6144), (10944, 5184, 512), (11712, 10624, 256), (11304, 720, 256), (9168, 18, 64), (6144, 4096, 8192)]                    # bf16 operands, f32 C
ALGOS_C32 = ["auto", "lp128", "lp256w4", "lp256m16", "stream64", "lp256x192"]
GRID_FP8 = [(2304, 8000, 128), (10688, 3136, 256), (8128, 9024, 512), (7424, 7680, 1024), (48448, 64, 256), (6976, 8448, 2048)]                      # e4m3 operands, bf16 C
ALGOS_FP8 = ["auto", "lp128", "lp256w4"]
GRID_TA = [(42320, 144, 6144), (37648, 192, 4096), (60544, 696, 6144), (744, 5432, 512), (10376, 64, 2048), (4096, 2048, 4096), (8192, 8192, 2048)]   # lhs stored [K][M] x row-major rhs
ALGOS_TA = ["auto", "lp128", "lp256w4"]
BAR_RATIO, BAR_US = 1.10, 2.0


def test_auto_is_within_10_percent_of_the_best_forced_kernel_on_every_shape_of_the_grid(client):
    audit(client, GRID, ALGOS, False, "select_audit.txt")


def test_auto_on_few_rows_times_a_row_major_weight(client):
    audit(client, GRID_NN, ALGOS_NN, True, "select_audit_nn.txt")


def test_auto_on_f32_few_rows(client):
    audit(client, GRID_F32, ALGOS_F32, False, "select_audit_f32.txt", f32=True)


def test_auto_on_f32_tiles_and_short_k(client):
    audit(client, GRID_F32_TILES, ALGOS_F32_TILES, False, "select_audit_f32_tiles.txt", f32=True)


def test_auto_with_an_f32_c(client):
    audit(client, GRID_C32, ALGOS_C32, False, "select_audit_c32.txt", c32=True)


def test_auto_on_fp8(client):
    audit(client, GRID_FP8, ALGOS_FP8, False, "select_audit_fp8.txt", fp8=True)


def test_auto_on_a_transposed_lhs(client):
    audit(client, GRID_TA, ALGOS_TA, True, "select_audit_ta.txt", ta=True)


def audit(client, grid, algos, nn, log_name, f32=False, **kind):
    sys.path.insert(0, str(ROOT / "tools"))
    sys.path.insert(0, str(ROOT))
    import ab_algos
    import bench
    ev = bench.Events(client)
    res = ab_algos.measure(client, ev, grid, algos, rounds=3, iters=10, nn=nn, f32=f32, **kind)

    def is_behind(r):
        us = {a: t for a, t in r["us"].items() if t == t}
        best = min(t for a, t in us.items() if a != "auto")
        return us["auto"] / best > BAR_RATIO and us["auto"] - best > BAR_US
    # a shape that looks behind is measured once more, longer, before it counts (a 20 us launch beside a DVFS step is noisy)
    suspects = [shape for shape, r in res.items() if is_behind(r)]
    if suspects:
        res.update(ab_algos.measure(client, ev, suspects, algos, rounds=7, iters=20, nn=nn, f32=f32, **kind))
    lines, behind = [], []
    for (m, n, k), r in res.items():
        us = {a: t for a, t in r["us"].items() if t == t}
        best_algo, best = min(((a, t) for a, t in us.items() if a != "auto"), key=lambda x: x[1])
        auto = us["auto"]
        ratio = auto / best
        # (AUTO behind the very kernel it selected is the measurement -- AUTO's turn comes first in every round, next to the clock's ramp: 42320 x 144 x 6144 137.5 us as
        #  AUTO -> lp256w4 against 124.8 forced in one evidence run of late round 6 -- not a selection)
        flag = ratio > BAR_RATIO and auto - best > BAR_US and best_algo != r["auto"]
        lines.append(f"{m}x{n}x{k}: AUTO -> {r['auto']:9s} {auto:8.1f} us   best forced {best_algo:9s} {best:8.1f} us   x{ratio:.3f}"
                     + ("   <-- BEHIND" if flag else "") + "   | " + "  ".join(f"{a} {t:.1f}" for a, t in us.items() if a != "auto"))
        if flag:
            behind.append(lines[-1])
    out = Path(os.environ.get("GRAFT_REPO_ROOT", ROOT)) / "gpurun_out"
    try:
        out.mkdir(exist_ok=True)
        (out / log_name).write_text("\n".join(lines) + "\n")
    except OSError:
        pass
    print("\n".join(lines))
    assert not behind, f"AUTO is more than {BAR_RATIO - 1:.0%} (and {BAR_US} us) behind a forced kernel:\n" + "\n".join(behind)
