"""tools/grid_model.py -- the replay that predicts the multi-GPU curve -- is pinned to the scheduler it claims to replay:
the trace recorder runs the real GridGp over a recording GridOps / GridComm, and its per-rank totals must be the counts the
scheduler itself reports (gpc_grid_stats) when the same problem is factored for real on the host stand-in (thread ranks).
The replay of the one-rank trace against the measured kernel costs must land on the measured one-rank grid run."""
import json
import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import grid_common as gc  # noqa: E402
import grid_model as gm  # noqa: E402
from gpc_amd import grid  # noqa: E402

COSTS = os.path.join(ROOT, "profiles", "r05_grid_costs.json")


@pytest.mark.parametrize("pr,pc,N,nb", [(2, 2, 1500, 128), (2, 4, 1500, 128), (4, 1, 1500, 128), (8, 1, 2300, 128), (3, 2, 1000, 128),
                                       (1, 4, 900, 128)])
def test_trace_totals_are_the_schedulers_own_counts(pr, pc, N, nb):
    hb = gc.host_binding()
    X, _, _ = gc.make_problem(N, 3, 1, 0, 5)
    grids = grid.create_local(pr, pc, nb, binding=hb)

    def work(g, rank):
        g.set_problem(gc.TERMS, X, None, None)
        g.stats(reset=True)
        assert g.update_k()[2] == 0
        return g.stats()

    try:
        real = grid.run_local(grids, work)
    finally:
        for g in grids:
            g.destroy()
    ops, traced = gm.trace(pr, pc, nb, N, 3)
    for r in range(pr * pc):
        for key in ("bytes_row", "bytes_col", "collectives", "update_flops", "update_launches", "update_bytes"):
            assert traced[r][key] == real[r][key], (r, key)
        mine = ops[r]
        upd = [o for o in mine if o["op"] == "update"]
        assert len(upd) == real[r]["update_launches"]
        assert abs(sum(o["flops"] for o in upd) - real[r]["update_flops"]) <= 1e-9 * max(real[r]["update_flops"], 1.0)
        # bytes this rank receives, from the exchanges the trace lists: a broadcast it is not the root of, the pieces of an
        # all-gather that are not its own
        got = {0: 0.0, 1: 0.0}
        for o in mine:
            if o["op"] == "bcast" and o["me"] != o["root"]:
                got[o["axis"]] += o["bytes"]
            elif o["op"] == "allgatherv":
                got[o["axis"]] += sum(p for i, p in enumerate(o["pieces"]) if i != o["me"])
        assert got[0] == real[r]["bytes_row"] and got[1] == real[r]["bytes_col"]
        n_exch = len([o for o in mine if o["op"] in ("bcast", "allgatherv")])
        # (the scheduler also counts the exchanges of groups of one, which never reach the transport)
        assert n_exch <= real[r]["collectives"]


def test_replay_of_one_rank_lands_on_the_measured_run():
    """cfg 3 through the grid path on ONE rank was measured (profiles/r05_bench_cfg3_grid_1x1.json); the replay of the 1 x 1
    trace against the measured kernel times has to reproduce it -- the model's only free parameters (link bandwidth,
    exchange latency) play no part here."""
    costs = gm.Costs(COSTS)
    one = gm.predict(costs, "cfg3", 1, 1, 1024, gm.Params())
    with open(os.path.join(ROOT, "profiles", "r05_bench_cfg3_grid_1x1.json")) as f:
        measured = json.loads(f.read().strip().splitlines()[-1])["ms_per_step"]
    assert abs(one["ms"] - measured) <= 0.04 * measured, (one["ms"], measured)


def test_predicted_curve():
    """What DESIGN.md's table says, recomputed: at 50 GB/s per link the tall layouts beat the wide ones, more ranks are
    faster, and 8 GPUs clear 6x of the one-rank replay on cfg 3."""
    costs = gm.Costs(COSTS)
    par = gm.Params(link_gbs=50.0)
    one = gm.predict(costs, "cfg3", 1, 1, 1024, par)["ms"]
    t = {s: gm.predict(costs, "cfg3", s[0], s[1], 1024, par)["ms"] for s in ((2, 1), (4, 1), (8, 1), (2, 4))}
    assert t[(8, 1)] < t[(4, 1)] < t[(2, 1)] < one
    assert t[(8, 1)] < t[(2, 4)]
    assert one / t[(8, 1)] >= 6.0
    # the order the scheduler issues by default (panel kernels before U2, exchanges beside it) beats the free-running one as
    # soon as a panel kernel cannot start beside a running update -- which is what MI355X does (tools/overlap_probe.py)
    free = gm.predict(costs, "cfg3", 8, 1, 1024, par, lookahead=2)["ms"]
    assert free > 1.1 * t[(8, 1)]
    # a slower link can only make it slower, and the ring form of the exchanges is no faster than the pairwise one
    slow = gm.predict(costs, "cfg3", 8, 1, 1024, gm.Params(link_gbs=25.0))["ms"]
    ring = gm.predict(costs, "cfg3", 8, 1, 1024, gm.Params(link_gbs=50.0, exchange="ring"))["ms"]
    assert slow >= t[(8, 1)] and ring >= t[(8, 1)]
