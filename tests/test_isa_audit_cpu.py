"""CPU: the load / wait order of the tick path's kernels, read from the compiler's own assembly (tools/isa_load_wait_audit.py).
Round 5 found the update kernel streaming with ONE 16-byte load per lane in flight -- each of its vector loads sat behind a per-lane
guard, in its own basic block with its own s_waitcnt vmcnt(0) -- and four more dependent-load chains on the small-K and blocking
paths (EXPERIMENTS.md 50-53).  These checks keep the rule DESIGN.md section 3 states: no global load behind a per-lane guard or a
barrier when its address is known earlier; and no scratch spills in the tick path's rollouts.  (Needs hipcc only: gfx950 cross-compiles.)"""
import importlib.util
import os
import re
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("c++filt") is None, reason="needs hipcc and c++filt")


def _tool():
    spec = importlib.util.spec_from_file_location("isa_audit", os.path.join(ROOT, "tools", "isa_load_wait_audit.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _longest_run_of_loads(seq):
    return max((len(x) for x in re.findall(r"L+", seq)), default=0)


def test_engine_kernels_issue_their_loads_before_waiting():
    k = _tool().audit(os.path.join(ROOT, "motion_planning_amd", "csrc", "mppi_engine.hip"))
    for name in ("mppi::update_kernel<float, true, 0, 8>", "mppi::update_kernel<double, true, 0, 8>", "mppi::update_kernel<float, true, 1, 8>",
                 "mppi::update_kernel<float, false, 0, 8>", "mppi::update_kernel<double, false, 0, 8>"):
        r = k[name]
        # the whole-chunk path: the row's eight HBM loads + the first group of per-sample totals in flight together; eight / seven
        # workgroups of 256 per CU (<= 72 VGPRs) and nothing in scratch
        assert _longest_run_of_loads(r["seq"]) >= 12, (name, r["seq"])
        # (the stored-noise instance of fp32 storage -- mppi_update / option store_eps, never a timed tick -- forms its candidates' fp64
        # weights inside the loop, next to the chunk it holds: 78 registers, six workgroups per CU)
        assert r["vgprs"] <= (80 if name == "mppi::update_kernel<float, false, 0, 8>" else 72) and r["scratch"] == 0, (name, r["vgprs"], r["scratch"])
    for name in ("mppi::update_kernel<float, true, 0, 16>", "mppi::update_kernel<double, true, 0, 16>"):
        assert _longest_run_of_loads(k[name]["seq"]) >= 12 and k[name]["scratch"] == 0, name
    # the merge kernel: four tuples per thread requested at once, in front of everything
    assert re.match(r"^L{16}", k["mppi::merge_kernel"]["seq"]), k["mppi::merge_kernel"]["seq"]
    # the merging publish kernel: four passes' tuples (64 rows) per round trip; the plain one: four words per thread
    assert _longest_run_of_loads(k["mppi::p2p_publish_merge_kernel"]["seq"]) >= 16
    assert _longest_run_of_loads(k["mppi::p2p_publish_kernel"]["seq"]) >= 4
    # the scan kernel reads its inputs once: no second trip to the (possibly pinned, PCIe) input slot in front of its first barrier
    for name in ("mppi::scan_tick_kernel<float, 1, true>", "mppi::scan_tick_kernel<double, 1, true>", "mppi::scan_tick_kernel<float, 4, true>"):
        head = k[name]["seq"].split("B")[0]
        assert head.count("L") <= 3, (name, head)   # the nominal controls' pair (+ the optional obstacle-grid lookup)


def test_lane_rollout_reads_nothing_behind_its_prologue_and_spills_nothing():
    k = _tool().audit(os.path.join(ROOT, "motion_planning_amd", "csrc", "rollout_f32_n4.hip"))
    for inline_nom in (1, 2):
        r = k["mppi::rollout_kernel<float, 4, true, false, %d, 0, false>" % inline_nom]
        # (the workgroup's first barrier = the end of the prologue; INLINE_NOM 2's scans have barriers of their own before it)
        behind = r["seq"].rsplit("B", 1)[1]
        assert "L" not in behind, (inline_nom, r["seq"][:200])
        assert r["scratch"] == 0 and r["spills"] == 0 and r["vgprs"] <= 96, (r["vgprs"], r["scratch"])
