#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric: SpartanSNARK::prove() wall-clock and R1CS constraints/s on the sha256_spartan circuit, 2 KiB message
(benches/sha256_spartan.rs:166-268: message vec![0u8; 2048], is_small = true, prove timed after warm-up proves on the same prep state). One prove
per "step"; the prep state (witness, cached Az/Bz/Cz, keys, matrices) is resident in HBM when the timed region starts.

    python bench.py --gpus N --steps K --warmup W [--workload c2|c3|c4|c5]

One process per GPU (torch.distributed.run for N > 1); the timed region is bracketed by barrier + torch.cuda.synchronize(), max over ranks.

  workload c2 (default)  every rank proves its own copy of the config-2 instance: N independent proofs, no data-path collective ("weak");
                         value = N * constraints / max-over-ranks time per step. The timed prove() does the reference's work: the transcript
                         prefix is re-hashed in every prove (the cached-prefix variant is reported beside it, never as `value`).
  workload c3            every rank proves its own batch of 32 Sha256StepCircuit instances + the core circuit through NeutronNovaZkSNARK::prove
                         (BASELINE config 3, benches/sha256_neutronnova.rs); "weak", no data-path collective;
  workload c5            NeutronNovaNIFS::prove over --instances (default 256) step instances of the 2 KiB SHA-256 shape (2^20 padded constraints
                         each: BASELINE config 5), instances sharded over the N ranks (nifs_prove_sharded: two field elements exchanged per round, one
                         bulk layer hand-off); the layers are prepared once as prep_prove does; "strong";
  workload c4            ONE proof of the synthetic 2^22 instance (BASELINE config 4, seed 0xDEADBEEF) sharded over the N ranks
                         (spartan2_amd/host/sharded_snark.cpp): rows/N Hyrax commitment, row-sliced Az/Bz/Cz, slice-sharded sum-checks with
                         one RCCL all-gather per round, column-sliced poly_ABC, point-range MSMs ("strong"); value = constraints / time.

Whatever the workload, the JSON line carries `sharded` — the config-4 legs on the N ranks of this run over an RCCL communicator of N ranks
(full-scalar Hyrax commit of 2^22 scalars = 2048 row MSMs of 2048 points sharded by row: MSM pairs/s; and the sharded 2^22 prove) — so that a
scaling run at N = 1, 2, 4, 8 holds north_star's MSM-throughput ratio. Rank 0 prints ONE JSON line with `roofline` (the bind kernel, HIP events
attached to its dispatches inside the timed region; `roofline_hbm` = the same kernel on 2^23-entry tables, past the 256 MiB Infinity Cache) and, at
N = 1, `cpu_baseline` (the CPU oracle's prove() of the same instance on the host cores, "port").
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch


def _pin_to_gpu_numa(dev):
    """Best effort: run this rank (and the threads and pinned buffers it creates from here on) on the CPUs local to its GPU's PCIe root
    (/sys/bus/pci/devices/<bdf>/local_cpulist). Every sum-check round is a host <-> device mailbox round trip; from the far socket of a
    two-socket node each one pays the inter-socket hop (measured: 1.395 vs 1.378 ms per prove). Returns the CPU count or None."""
    try:
        p = torch.cuda.get_device_properties(dev)
        bdf = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
        with open(f"/sys/bus/pci/devices/{bdf}/local_cpulist") as f:
            spec = f.read().strip()
        cpus = set()
        for part in spec.split(","):
            if "-" in part:
                a, b = part.split("-")
                cpus.update(range(int(a), int(b) + 1))
            elif part:
                cpus.add(int(part))
        cpus &= os.sched_getaffinity(0)
        if len(cpus) < 4:
            return None
        os.sched_setaffinity(0, cpus)
        return len(cpus)
    except Exception:
        return None


KERNEL_CLASSES = ("bind_stream_cubic", "bind_stream_quad", "bind_stream_quad_sparse", "bind", "eval_cubic", "eval_quad", "spmv_incremental", "round0_products", "poly_abc", "eq_table",
                  "rowmat_vec", "msm_sort", "msm_bucket_sum", "msm_window_reduce", "fixed_base")


def _c4_instance():
    from spartan2_amd import frontend

    return frontend.synthetic_circuit(45000, 0xDEADBEEF, num_public=8)  # N = M = 2^22, SHA-like row mix, Bernoulli(1/2) bits


_GUARDIAN = r"""
import signal, sys
for s in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP):
    signal.signal(s, signal.SIG_IGN)  # the launcher ends the ranks when one of them dies: this process outlives them by one write
last, done = None, False
for ln in sys.stdin:
    ln = ln.rstrip("\n")
    if ln == "done":
        done = True
        break
    if ln:
        last = ln
if not done and last is not None:
    sys.stdout.write(last + "\n")
    sys.stdout.flush()
"""


class LineGuardian:
    """Rank 0's bench line, kept by a small child process while the sharded legs run: a watchdog covers a leg that never returns, not a process that
    DIES in one (a signal inside native collective code that has never run on this pool; the launcher's SIGTERM after another rank died). The child
    (a fresh interpreter: no GPU state) holds the latest snapshot of the line — the headline and the legs finished so far — and prints it if its input ends
    without the closing word; on a normal end it prints nothing."""

    def __init__(self):
        import subprocess

        sys.stdout.flush()
        self.p = subprocess.Popen([sys.executable, "-c", _GUARDIAN], stdin=subprocess.PIPE, text=True)

    def snapshot(self, line, legs, name):
        marked = dict(legs)
        marked[name] = {"error": "the process ended inside this leg; the line was printed by its guardian with the legs finished before it"}
        snap = dict(line)
        snap["sharded"] = marked
        try:
            self.p.stdin.write(json.dumps(snap) + "\n")
            self.p.stdin.flush()
        except Exception:
            pass

    def done(self):
        try:
            self.p.stdin.write("done\n")
            self.p.stdin.close()
            self.p.wait(timeout=10)
        except Exception:
            pass


class LegDog:
    """One watchdog per extra leg: arm(name) before a leg, disarm() after the last. A leg that outlives its allowance ends the process - rank 0 first
    prints the bench line it has (`line`, whose "sharded" object holds the legs that finished) with the leg marked as not finished."""

    def __init__(self, rank, line, legs, seconds, guardian=None):
        self.rank, self.line, self.legs, self.seconds, self.timer, self.guardian = rank, line, legs, seconds, None, guardian

    def _expired(self, name):
        if self.rank == 0 and self.line is not None:
            if self.guardian:
                self.guardian.done()
            self.legs[name] = {"error": f"did not finish within {self.seconds:.0f} s; the line is printed with the legs that did"}
            print(json.dumps(self.line), flush=True)
        os._exit(0)

    def arm(self, name):
        import threading

        self.disarm()
        if self.guardian and self.line is not None:
            self.guardian.snapshot(self.line, self.legs, name)
        if os.environ.get("SPARTAN_BENCH_DIE_IN") == name:  # test switch of this script (tests/test_gpu_bench_rehearsal.py): what a crash in native code does
            import signal

            os.kill(os.getpid(), signal.SIGKILL)
        self.timer = threading.Timer(self.seconds, self._expired, args=(name,))
        self.timer.daemon = True
        self.timer.start()

    def disarm(self):
        if self.timer is not None:
            self.timer.cancel()
            self.timer = None


def _leg_done(rank, name, leg):
    if rank == 0:  # progress on stderr as each leg lands (stdout carries the one JSON line)
        print(f"[bench] sharded.{name}: {json.dumps(leg)}", file=sys.stderr, flush=True)


def sharded_legs(ctx, comm, group, steps_prove, steps_commit, check_oracle, out=None, dog=None):
    """BASELINE config 4 on the ranks of this run: (1) PCS::commit of 2^22 full-width scalars, rows sharded by row - first, before any prove-side
    collective; (1b) one general 2^20-point MSM sharded by point range; (2) one sharded prove. `out` is filled leg by leg (a watchdog that fires in a
    later leg still has the earlier ones); `dog` is re-armed before each."""
    from spartan2_amd import hip, host

    rank, world = comm.rank, comm.world
    out = {} if out is None else out
    out.update({"rccl_ranks": world, "exchange_backend": comm.backend})
    if dog:
        dog.arm("c4_commit")
    # ---- (1) MSM leg: 2048 row MSMs of 2048 points, rows / world per rank, one all-gather of 64-byte rows
    g = host.from_label(b"ck", 2049)
    key = hip.CommitmentKey(ctx, g[:2048], g[2048])
    rows_local = 2048 // world
    rng = np.random.default_rng(0xC4 + rank)
    n_local = rows_local * 2048
    v = rng.integers(0, 1 << 63, size=(n_local, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n_local, 4), dtype=np.uint64)
    v[:, 3] &= np.uint64((1 << 63) - 1)
    blinds = rng.integers(0, 1 << 62, size=(rows_local, 4), dtype=np.uint64)
    t = hip.Table.from_host(ctx, v)
    t_w = time.perf_counter()
    host.sharded_commit(ctx, comm, key, t, n_local, blinds)  # warm-up: builds this rank's comb table of the key (every rank holds the whole key's table)
    t_first = time.perf_counter() - t_w
    group.barrier()
    t0 = time.perf_counter()
    for _ in range(steps_commit):
        rows = host.sharded_commit(ctx, comm, key, t, n_local, blinds)
    group.barrier()
    dt = group.max_over_ranks(time.perf_counter() - t0) / steps_commit
    t.free()
    comb_bits = int(os.environ.get("SPARTAN_COMB_BITS", "13")) or 13  # capi_comb.hip's default
    comb_windows = -(-257 // comb_bits)
    out["c4_commit"] = {"scalars": 1 << 22, "rows": 2048, "rows_per_rank": rows_local, "ms": dt * 1e3, "msm_pairs_per_s": (1 << 22) / dt,
                        "ec_additions_per_s": (1 << 22) * comb_windows / dt,
                        "note": "2048 x 2048 full-width scalars over one key (hyrax_pc.rs:230-300), rows sharded by row, fixed-base comb table of the key with "
                                f"{comb_bits}-bit signed windows: {comb_windows} mixed additions per (scalar, base) pair (the bucket form of round 1 needed ~36)"}
    # the data path's own exchange at this world size: small record (a round's sums) and the commitment rows of one rank
    ex_small, ex_rows = [], []
    try:
        rec = np.zeros((1, 12), dtype=np.uint64)
        rows_rec = np.zeros((rows_local, 8), dtype=np.uint64)
        for _ in range(3):
            comm.allgather(rec)
        for _ in range(20):
            t1 = time.perf_counter()
            comm.allgather(rec)
            ex_small.append(time.perf_counter() - t1)
        for _ in range(5):
            t1 = time.perf_counter()
            comm.allgather(rows_rec)
            ex_rows.append(time.perf_counter() - t1)
    except Exception as exc:
        out["c4_commit"]["exchange_error"] = repr(exc)
    if ex_small:
        per = out["c4_commit"]["ms"]
        small_us, rows_us = sorted(ex_small)[len(ex_small) // 2] * 1e6, sorted(ex_rows)[len(ex_rows) // 2] * 1e6
        out["c4_commit"].update({
            "first_call_ms_incl_comb_table_build": t_first * 1e3, "comb_table_build_ms_per_rank": max(0.0, t_first * 1e3 - per),
            "exchange_96B_us_median": small_us, "exchange_rows_block_us_median": rows_us,
            "prediction_for_the_first_multi_gpu_run": {
                "basis": "measured at THIS world size: the commit is row-parallel with no reduction (each rank commits rows / N rows over its own comb table, one "
                         "all-gather of 64-byte rows), so t(N) = t(1) * (rows/N)/rows + exchange(rows/N rows); xGMI all-gather of 128 KiB total is taken at the "
                         "measured single-rank exchange cost plus 10 us per extra rank (ring steps over ~153 GB/s links: bandwidth is negligible at this size)",
                "c4_commit_ms": {str(n): (per * world / n) + (rows_us + 10.0 * (n - 1)) / 1e3 for n in (1, 2, 4, 8)},
                "c4_commit_speedup_vs_1": {str(n): (per * world) / ((per * world / n) + (rows_us + 10.0 * (n - 1)) / 1e3) for n in (2, 4, 8)},
                "north_star": ">= 6x MSM throughput at 8 GPUs vs 1"}})
    _leg_done(rank, "c4_commit", out["c4_commit"])
    if dog:
        dog.arm("msm_general")
    # ---- (1b) one general Pippenger MSM of 2^20 caller-supplied points (no precomputed tables), sharded by POINT RANGE: every rank runs the
    # multi-block Pippenger on its range of the device-resident operands, the affine partial sums are gathered and added
    try:
        nbig = 1 << 20
        prng = np.random.default_rng(0xB16)  # the same operands on every rank
        tt = prng.integers(0, 1 << 62, size=(nbig, 4), dtype=np.uint64)
        pts = np.concatenate([key.fixed_base_mul_h(tt[lo:lo + (1 << 16)]) for lo in range(0, nbig, 1 << 16)])  # t_i * h: distinct points
        sc = prng.integers(0, 1 << 63, size=(nbig, 4), dtype=np.uint64) * np.uint64(2) + prng.integers(0, 2, size=(nbig, 4), dtype=np.uint64)
        sc[:, 3] &= np.uint64((1 << 63) - 1)
        dev, tab = hip.Points(ctx, pts), hip.Table.from_host(ctx, sc)
        lo, hi = nbig * rank // world, nbig * (rank + 1) // world

        def big():
            part = hip.msm_points(ctx, tab, lo, hi - lo, dev, lo)
            return hip.point_sum(comm.allgather(part.reshape(1, 8)).reshape(world, 8)) if world > 1 else part

        first = big()
        group.barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            res = big()
        group.barrier()
        dt = group.max_over_ranks(time.perf_counter() - t0) / 3
        pw = int(hip.lib().sp_msm_pippenger_window(ctypes.c_size_t(hi - lo)))
        nwin = -(-257 // pw)
        out["msm_general"] = {"points": nbig, "points_per_rank": hi - lo, "ms": dt * 1e3, "msm_pairs_per_s": nbig / dt, "ec_additions_per_s": nbig * nwin / dt,
                              "window_bits": pw, "deterministic": bool((first == res).all()),
                              "note": "DlogGroupExt::vartime_multiscalar_mul (msm.rs:187-222) on caller-supplied bases resident in HBM: signed-digit Pippenger, multi-block "
                                      "counting sort, bucket lists cut into tasks, bit-sliced window sums (kernels_pippenger.hpp); no per-base tables"}
        tab.free()
        dev.free()
    except Exception as exc:  # the leg must not take the commit / prove numbers with it
        out["msm_general"] = {"error": repr(exc)}
    _leg_done(rank, "msm_general", out["msm_general"])
    if dog:
        dog.arm("c4_prove")
    # ---- (2) one proof of the 2^22 instance over all ranks
    inst = _c4_instance()
    t0 = time.time()
    sn = host.ShardedSpartanSNARK(ctx, comm, inst)
    t_setup = time.time() - t0
    tape = np.random.default_rng(0xDEADBEEF).integers(0, 256, size=(8192, 64), dtype=np.uint8)
    used = sn.prep_prove(tape)
    step_tape = np.random.default_rng(0xDEADBEF0).integers(0, 256, size=(8192, 64), dtype=np.uint8)
    words, _, _ = sn.prove(step_tape)
    ex0 = comm.stats()["exchanges"]
    group.barrier()
    t0 = time.perf_counter()
    acc = {}
    for _ in range(steps_prove):
        words, _, ph = sn.prove(step_tape)
        for k_, v_ in ph.items():
            acc[k_] = acc.get(k_, 0.0) + v_
    group.barrier()
    dt = group.max_over_ranks(time.perf_counter() - t0) / steps_prove
    out["c4_prove"] = {"num_cons_unpadded": inst.num_cons, "num_cons": sn.dims["num_cons"], "ms": dt * 1e3, "constraints_per_s": inst.num_cons / dt,
                       "exchanges_per_prove": (comm.stats()["exchanges"] - ex0) / steps_prove, "setup_s": t_setup,
                       "phases_ms": {k_: v_ / steps_prove for k_, v_ in acc.items() if k_ != "exchanges"}}
    if check_oracle and rank == 0:
        import oracle_lib as ol  # test infrastructure: the checker only

        osp = ol.OracleSpartan(inst)
        assert osp.prep_prove(tape) == used
        want, _, secs = osp.prove(step_tape)
        out["c4_prove"]["bit_exact_vs_cpu_oracle"] = bool((want == words).all())
        out["c4_prove"]["cpu_oracle_ms"] = secs * 1e3
    _leg_done(rank, "c4_prove", out["c4_prove"])
    sn.close()
    return out


def _scaling_vs_1(world, args, out):
    """Raw ratios against the N = 1 line of the same command on the same box, when one is cached (a run at N = 1 writes it; the driver runs N = 1, 2, 4, 8
    back to back). Ratios of measured figures only, no efficiency claim: the driver computes that itself from the per-N values."""
    import tempfile

    # a directory of this user's own (0700, ownership checked), not a predictable file name in the shared temp directory
    cache = os.path.join(tempfile.gettempdir(), f"spartan2_amd_bench_{os.getuid()}")
    try:
        os.makedirs(cache, mode=0o700, exist_ok=True)
        st = os.lstat(cache)
        import stat as _stat

        if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
            return None
    except OSError:
        return None
    path = os.path.join(cache, f"n1_{args.workload}_{args.message_bytes}.json")
    legs = out.get("sharded") or {}
    mine = {"value": out["value"], "ms_per_step": out["ms_per_step"],
            "msm_pairs_per_s": (legs.get("c4_commit") or {}).get("msm_pairs_per_s"), "c4_prove_ms": (legs.get("c4_prove") or {}).get("ms")}
    try:
        if world == 1:
            with open(path, "w") as f:
                json.dump(mine, f)
            return None
        with open(path) as f:
            one = json.load(f)
    except OSError:
        return None
    ratio = lambda a, b: (a / b) if (a and b) else None
    return {"source": path, "n1": one, "value_ratio": ratio(mine["value"], one["value"]),
            "msm_pairs_per_s_ratio": ratio(mine["msm_pairs_per_s"], one["msm_pairs_per_s"]),
            "c4_prove_speedup": ratio(one["c4_prove_ms"], mine["c4_prove_ms"])}


def _self_launch(n):
    """`python bench.py --gpus N` without a launcher around it: become `python -m torch.distributed.run --standalone --local-addr 127.0.0.1
    --nnodes=1 --nproc-per-node N bench.py <same flags>` (one rank per GPU; the ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).
    The reference's harness takes its thread count the same way, as a parameter of ONE binary (benches/sha256_spartan.rs:155-164,192-199)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL between processes fails without it on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    # --standalone: the launcher's own rendezvous store binds a free port itself (nothing is picked here and released for someone else to take)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={n}",
           os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(cmd[0], cmd, env)


def dist(ms):
    """min / median / max of the per-step times of rank 0 (ms): ms_per_step is the mean the contract asks for, this shows what it is made of"""
    a = sorted(ms)
    if not a:
        return None
    med = a[len(a) // 2]
    return {"min": a[0], "median": med, "p90": a[(9 * len(a)) // 10], "max": a[-1], "over_2x_median": sum(1 for x in a if x > 2 * med),
            "steps_over_2x_median": [i for i, x in enumerate(ms) if x > 2 * med][:16], "n": len(a)}


def prep_prove_leg(snark, inst, tape, step_tape, words, calls=6, oracle=None, used=None):
    """SpartanSNARK::prep_prove as a phase of its own, the protocol of benches/sha256_spartan.rs:206-222: setup outside, a FRESH prep_prove inside the timed
    routine, its output dropped outside (Criterion's iter_batched). min / median over `calls` calls of the C entry point, the driver's own phases
    (witness upload + expansion on the device, Hyrax commit of the precommitted rows, the queueing of the row tables, multiply_vec_precommitted, scratch), the
    first prove on each fresh state, the same with the FixedBaseMul tables of the committed rows built and waited for INSIDE prep_prove (round 5's
    behaviour; SPARTAN_PREP_TABLES=sync), and the CPU oracle's prep_prove of the same instance beside it. Every prove on every state is the timed proof."""
    import ctypes

    from spartan2_amd import host as _host

    names = ["witness", "commit", "tables", "matvec", "scratch", None, "total", None]

    def run(mode, n):
        prev = os.environ.get("SPARTAN_PREP_TABLES")
        if mode:
            os.environ["SPARTAN_PREP_TABLES"] = mode
        elif prev is not None:
            del os.environ["SPARTAN_PREP_TABLES"]
        rows, same = [], True
        try:
            for _ in range(n):
                tf = time.perf_counter()
                if snark.ps:
                    _host.lib().ss_prep_free(snark.ps)
                    snark.ps = None
                t0 = time.perf_counter()
                u = snark.prep_prove(tape)
                t1 = time.perf_counter()
                assert used is None or u == used
                ms = (ctypes.c_double * 8)()
                _host.lib().ss_prep_phases(snark.ps, ms)
                snark.set_flags(prefix_cache=False)
                w1, _, _ = snark.prove(step_tape)
                t2 = time.perf_counter()
                same = same and bool((np.asarray(w1) == np.asarray(words)).all())
                rows.append({"prep_ms": (t1 - t0) * 1e3, "first_prove_ms": (t2 - t1) * 1e3, "drop_previous_ms": (t0 - tf) * 1e3,
                             "phases_ms": {k: v for k, v in zip(names, ms) if k}})
        finally:
            if prev is None:
                os.environ.pop("SPARTAN_PREP_TABLES", None)
            else:
                os.environ["SPARTAN_PREP_TABLES"] = prev
        rows = rows[1:] if len(rows) > 2 else rows  # (the first call of a mode also pays one-off allocations of the context's workspaces)
        pm = sorted(r["prep_ms"] for r in rows)
        med = rows[[r["prep_ms"] for r in rows].index(pm[len(pm) // 2])]
        return {"calls": len(rows), "min_ms": pm[0], "median_ms": pm[len(pm) // 2], "max_ms": pm[-1], "median_call_phases_ms": med["phases_ms"],
                "first_prove_after_ms_median": sorted(r["first_prove_ms"] for r in rows)[len(rows) // 2],
                "drop_of_previous_state_ms_median": sorted(r["drop_previous_ms"] for r in rows)[len(rows) // 2], "proofs_identical_to_the_timed_one": same}

    leg = {"protocol": "fresh prep_prove per call on one key, previous state dropped outside the timed call (benches/sha256_spartan.rs:206-222); inputs are the "
                       "witness as machine words in pageable host memory (8 B a value: the PCIe transfer is inside the figure)",
           "default": run(None, calls + 1), "tables_built_inside_prep": run("sync", 4), "no_tables": run("off", 4),
           "note": "default = the row tables (FixedBaseMul of the committed rows, for comm_LZ) are queued behind the FIRST prove on a state and used from "
                   "the first prove that finds them built; tables_built_inside_prep = round 5's behaviour with round 6's build (three launches, batch inversion)"}
    if oracle is not None:
        t0 = time.perf_counter()
        ou = oracle.prep_prove(tape)
        leg["cpu_oracle_prep_prove_ms"] = (time.perf_counter() - t0) * 1e3
        assert used is None or ou == used
    return leg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=("c2", "c3", "c4", "c5"), default="c2")
    ap.add_argument("--instances", type=int, default=256, help="workload c5: step instances in the batch (a power of two, >= 2 per rank)")
    ap.add_argument("--message-bytes", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sharded", action="store_true", help="skip the config-4 sharded legs (profiling runs)")
    ap.add_argument("--extras-timeout", type=float, default=600.0, help="seconds the sharded extra legs of the default workload may take before the line is printed without them")
    ap.add_argument("--concurrent", type=int, default=8, help="extra (untimed) leg: this many independent proofs in flight on the one GPU; 0 = skip")
    args = ap.parse_args()

    from spartan2_amd import dist as spd

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _self_launch(args.gpus)  # does not return: this process becomes the launcher of the N ranks
    rank, local_rank, world = spd.env_rank()
    # Rehearsal of the N > 1 control flow on a box with ONE GPU (SPARTAN_BENCH_REHEARSAL=1; never set by the driver): every rank uses device 0,
    # torch.distributed runs over gloo and the data-path exchange goes through the callback backend (RCCL refuses two ranks on one device).
    # Its numbers mean nothing (the ranks share the GPU); the line says so. What it exercises is everything between the launcher and the JSON line.
    rehearsal = os.environ.get("SPARTAN_BENCH_REHEARSAL") == "1" and world > 1
    if rehearsal:
        local_rank = 0
    if world != args.gpus:
        raise SystemExit(f"rank {rank}: --gpus {args.gpus} but WORLD_SIZE={world} (the launcher's --nproc-per-node must equal --gpus)")
    if not torch.cuda.is_available():
        raise SystemExit(f"rank {rank}: bench.py needs {world} MI355X device(s), this node shows none: libspartan_hip has no CPU fallback")
    if torch.cuda.device_count() < world and not rehearsal:
        raise SystemExit(f"rank {rank}: --gpus {world} needs {world} devices (one process per GPU), this node shows {torch.cuda.device_count()}")
    torch.cuda.set_device(local_rank)
    numa_cpus = _pin_to_gpu_numa(local_rank)  # before any pinned allocation or helper thread exists
    comm_backend = "torch" if rehearsal else "rccl"
    group = spd.Group(backend="gloo" if rehearsal else spd.RCCL_BACKEND)  # RCCL: barrier and max-over-ranks of the timed region, and the hand-over of the C++ communicator's id

    from spartan2_amd import frontend, hip, host

    ctx = hip.Context(local_rank)
    # the data-path exchange layer (ncclAllGather from C++ on a communicator of its own): created here for the workloads that ARE the sharded path; the
    # default workload creates it only for its extra legs, after the timed region and under a watchdog (see below)
    comm = host.Comm(rank, world, comm_backend, device=local_rank) if args.workload in ("c4", "c5") else None
    barrier = group.barrier  # dist.barrier() + torch.cuda.synchronize()

    if args.workload == "c3":
        # ---- 32 step circuits + core through NeutronNovaZkSNARK::prove on every rank (weak scaling, independent batches)
        nsteps = 32
        circs = [frontend.sha256_step_circuit(bytes([(i + 37 * rank) & 0xFF]) * 64) for i in range(nsteps)]
        core = frontend.sha256_step_circuit(bytes(64))
        nn = host.NeutronNovaZkSNARK(ctx, circs, core)
        tape = np.random.default_rng(0xC3 + rank).integers(0, 256, size=(32768, 64), dtype=np.uint8)
        used = nn.prep_prove(tape)
        step_tape = tape[used:]
        first_words, first_used, _ = nn.prove(step_tape)  # the proof the oracle must reproduce (every prove rerandomizes the prep state in place)
        for _ in range(args.warmup):
            words, _, _ = nn.prove(step_tape)
        barrier()
        t0 = time.perf_counter()
        acc = {}
        c3_step_ms, c3_slowest = [], None
        for _ in range(args.steps):
            ts = time.perf_counter()
            words, _, ph = nn.prove(step_tape)
            c3_step_ms.append((time.perf_counter() - ts) * 1e3)
            if c3_step_ms[-1] >= max(c3_step_ms):
                c3_slowest = dict(ph)
            for k_, v_ in ph.items():
                acc[k_] = acc.get(k_, 0.0) + v_
        barrier()
        elapsed = group.max_over_ranks(time.perf_counter() - t0)
        ncons = circs[0].num_cons * nsteps + core.num_cons
        # the reference-order driver (one thread, statement order of src/neutronnova_zk.rs:1609-2093, PCS::prove as one sp_hyrax_prove) on a prep state of its
        # own: its first proof must be the default driver's first proof (every prove rerandomizes its prep state in place), then the same number of steps
        nn_ref = host.NeutronNovaZkSNARK(ctx, circs, core)
        assert nn_ref.prep_prove(tape) == used
        ref_first, _, _ = nn_ref.prove(step_tape, reference_order=True)
        ref_identical = bool(len(ref_first) == len(first_words) and (ref_first == first_words).all())
        for _ in range(args.warmup):
            nn_ref.prove(step_tape, reference_order=True)
        barrier()
        t0 = time.perf_counter()
        acc_ref = {}
        for _ in range(args.steps):
            _, _, ph = nn_ref.prove(step_tape, reference_order=True)
            for k_, v_ in ph.items():
                acc_ref[k_] = acc_ref.get(k_, 0.0) + v_
        barrier()
        elapsed_ref = group.max_over_ranks(time.perf_counter() - t0)
        nn_ref.close()
        # untimed pass with the kernel classes instrumented: the streaming kernels of the NIFS rounds against the HBM roofline (at 32 x 2^15 every launch
        # is latency-bound: the numbers say how far a 1 MiB-per-layer launch is from the rate the same kernels reach at config 5's size)
        c3_names = ("nifs_fold_prove", "nifs_round0_small", "nifs_round0", "nifs_fold", "nifs_cvals", "fold_tables", "eval_cubic_pow", "eval_quad", "bind", "poly_abc", "fixed_base")
        ctx.reset_stats(True)
        ctx.stats_filter("")
        nbp = 3
        for _ in range(nbp):
            nn.prove(step_tape)
        c3_stats = {k_: ctx.kernel_stats(k_) for k_ in c3_names}
        ctx.reset_stats(False)
        # NeutronNovaZkSNARK::verify on the device-backed driver (outside the timed region): the last proof of the loop
        verify_rc = nn.verify(words)
        tv = time.perf_counter()
        for _ in range(5):
            verify_rc |= nn.verify(words)
        verify_ms = (time.perf_counter() - tv) / 5 * 1e3
        if verify_rc != 0:
            raise SystemExit(f"the device-backed verifier rejected the proof (check {verify_rc})")
        if rank == 0:
            out = {"metric": "sha256_neutronnova 32 step circuits, NeutronNovaZkSNARK::prove: R1CS constraints/sec (all step + core constraints of one batch per prove)",
                   "value": ncons * world * args.steps / elapsed, "unit": "constraints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                   "dtype": "u256 modular integer (8 x u32 Montgomery limbs; T256 scalar/base fields)",
                   "data": "synthetic: Sha256StepCircuit over constant 64-byte blocks, seeded randomness tape",
                   "config": {"workload": "sha256_neutronnova 32 step circuits (BASELINE config 3), NeutronNovaZkSNARK::prove", "num_steps": nsteps,
                              "num_cons_unpadded_per_step": circs[0].num_cons, "num_cons_per_step": 1 << nn.info["nx"],
                              "parallelism": f"{world} independent batches, one per GPU",
                              "host_walkers": int(hip.lib().sp_walkers()),
                              "host_walkers_note": "polling host threads of the process (SPARTAN_WALKERS): the round commitments of the verifier circuit "
                                                   "(sp_hyrax_commit_split_*) and the host loops of its instance are spread over them; 0 = the device walk, one host thread"},
                   "phases_ms": {k_: v_ / args.steps for k_, v_ in acc.items()}, "step_ms_distribution": dict(dist(c3_step_ms), slowest_step_phases_ms=c3_slowest),
                   "verify_ms": verify_ms, "sharded": None, "roofline": None, "cpu_baseline": None,
                   "reference_order": {"ms_per_step": elapsed_ref / args.steps * 1e3, "constraints_per_s": ncons * world * args.steps / elapsed_ref,
                                       "proof_identical": ref_identical, "phases_ms": {k_: v_ / args.steps for k_, v_ in acc_ref.items()},
                                       "headline_over_reference_order": elapsed / elapsed_ref,
                                       "note": "one thread, statement order of src/neutronnova_zk.rs:1609-2093, ABI calls only (no jobs on a second context, PCS::prove = one "
                                               "sp_hyrax_prove): the time of an unchanged neutronnova_zk.rs over the ABI"}}
            fp = c3_stats["nifs_fold_prove"]
            if fp[1]:
                ach = (fp[2] / fp[1]) / (fp[0] / fp[1] * 1e-3) / 1e9
                out["roofline"] = {"bound": "hbm", "kernel": "k_nifs_fold_prove (merged fold + prove rounds of NeutronNovaNIFS::prove: 384 B per k and prove pair, SURVEY 8(d))",
                                   "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": None, "launches": fp[1] / nbp,
                                   "avg_launch_us": fp[0] / fp[1] * 1e3, "alg_bytes_per_launch": fp[2] / fp[1],
                                   "note": "HIP events of an untimed instrumented pass; at 32 instances x 2^15 constraints a launch moves ~12 MB and is latency-bound",
                                   "other_kernels": {k_: {"launches_per_step": v_[1] / nbp, "avg_us": v_[0] / max(v_[1], 1) * 1e3,
                                                          "alg_GBps": (v_[2] / max(v_[0], 1e-9)) / 1e6} for k_, v_ in c3_stats.items() if v_[1] and k_ != "nifs_fold_prove"}}
            if world == 1 and not args.no_cpu_baseline:
                import oracle_lib as ol  # test infrastructure, used here only as the reported CPU baseline and the bit-exactness check

                cores = ol.lib().orc_set_threads(min(spd.cpu_budget(), 32))  # OpenMP threads past the cgroup's CPU quota only get the whole process throttled
                onn = ol.OracleNeutronNova(circs, core)
                want, oused, secs = onn.prove(tape)
                ok = bool(oused[0] == used and oused[1] == first_used and len(want) == len(first_words) and (want == first_words).all()) and onn.verify_words(first_words) == 0
                out["cpu_baseline"] = {"value": ncons / secs, "unit": "constraints/s", "cores": cores, "kind": "port",
                                       "sample": f"one NeutronNovaZkSNARK::prove of the same 32 + 1 circuits on the CPU oracle (C++ restatement, OpenMP over {cores} threads): "
                                                 f"{secs * 1e3:.0f} ms (prep_prove not included)",
                                       "ms": secs * 1e3, "gpu_proof_bit_exact_and_verified": ok}
                if not ok:
                    raise SystemExit("GPU proof differs from the oracle's or fails verification")
            print(json.dumps(out))
        nn.close()
        ctx.close()
        group.close()
        return

    if args.workload == "c5":
        # ---- NeutronNova NIFS at config 5's size: 256 instances x 2^20 constraints, sharded by instance over the ranks
        n_total = args.instances
        n_local = n_total // world
        if n_local < 2 or n_local * world != n_total or n_total & (n_total - 1):
            raise SystemExit("--instances must be a power of two with at least two instances per rank")
        CW = 2048  # DEFAULT_COMMITMENT_WIDTH (src/provider/pcs/hyrax_pc.rs)
        distinct = [frontend.sha256_circuit(bytes([b]) * args.message_bytes) for b in (0, 1)]  # two witnesses, cycled over the batch
        mats, dims = host.pad_shape(distinct[0])
        shape = hip.Shape(ctx, mats, dims)
        nv = dims["num_shared"] + dims["num_precommitted"] + dims["num_rest"]
        rows = nv // CW
        g = host.from_label(b"ck", CW + 1)
        key = hip.CommitmentKey(ctx, g[:CW], g[CW])
        src = [hip.Table.from_host(ctx, host.padded_witness_limbs(dims, c.witness)) for c in distinct]
        X2 = [host.mont_limbs_from_u64(c.publics) for c in distinct]
        rng = np.random.default_rng(0xC5 + rank)
        tabs, comms, Xs, rWs = [], [], [], []
        for i in range(n_local):
            gi = rank * n_local + i
            t = hip.Table.zeros(ctx, nv)
            t.copy_from(0, src[gi & 1], 0, nv)
            bl = rng.integers(0, 1 << 62, size=(rows, 4), dtype=np.uint64)
            tabs.append(t)
            comms.append(key.commit(t, 0, nv, bl, True))
            Xs.append(X2[gi & 1])
            rWs.append(bl)
        comms, Xs, rWs = np.stack(comms), np.stack(Xs), np.stack(rWs)
        prepared = host.nifs_prepare(ctx, shape, dims, Xs, tabs, True)
        out_tabs = dict(A=hip.Table.zeros(ctx, dims["num_cons"]), B=hip.Table.zeros(ctx, dims["num_cons"]), C=hip.Table.zeros(ctx, dims["num_cons"]),
                        folded_W=hip.Table.zeros(ctx, nv))

        def one():
            tr = hip.Transcript(ctx, b"neutronnova_prove")
            vc = hip.Transcript(ctx, b"vc")  # stand-in for process_round: absorb the round polynomial, squeeze the challenge

            def hook(t, co):
                vc.absorb(b"p", np.ascontiguousarray(co, dtype=np.uint64).tobytes())
                return vc.squeeze(b"c")

            return host.nifs_prove_sharded(ctx, comm, shape, dims, key, comms, Xs, tabs, rWs, True, tr, hook, prepared=prepared, out_tabs=out_tabs)

        for _ in range(args.warmup):
            out = one()
        ctx.reset_stats(True)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            out = one()
        ctx.synchronize()
        barrier()
        elapsed = group.max_over_ranks(time.perf_counter() - t0)
        names = ("nifs_round0_small", "nifs_round0", "nifs_fold_prove", "nifs_prove_pairs", "nifs_fold", "nifs_cvals", "fold_tables")
        ks = {k_: ctx.kernel_stats(k_) for k_ in names}
        ctx.reset_stats(False)
        if rank == 0:
            ncons = distinct[0].num_cons * n_total
            res = {"metric": "NeutronNovaNIFS::prove over the step instances of one batch (layers prepared as prep_prove does): R1CS constraints folded per second",
                   "value": ncons * args.steps / elapsed, "unit": "constraints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                   "dtype": "u256 modular integer (8 x u32 Montgomery limbs); round 0 on the i64 mirrors (small-value mode)",
                   "data": "synthetic: two distinct SHA-256 witnesses of the 2 KiB shape cycled over the batch, seeded blinds; process_round replaced by a plain transcript",
                   "config": {"workload": f"NeutronNova NIFS, {n_total} step instances x 2^{dims['num_cons'].bit_length() - 1} constraints (BASELINE config 5), sharded by instance",
                              "instances": n_total, "instances_per_rank": n_local, "num_cons_unpadded": distinct[0].num_cons, "num_cons": dims["num_cons"],
                              "parallelism": f"{n_local} instances per rank over {world} ranks; 2 field elements exchanged per round, one layer hand-off",
                              "rccl_ranks": world, "exchanges_per_prove": comm.stats()["exchanges"] / (args.steps + args.warmup),
                              "resident_GB_per_rank": round(((4.5 * 32 + 3 * 8) * dims["num_cons"] + 32 * nv) * n_local / 1e9, 1)},
                   "kernel_ms_per_step": {k_: v_[0] / args.steps for k_, v_ in ks.items() if v_[1]},
                   "kernel_alg_GBps": {k_: v_[2] / max(v_[0], 1e-9) / 1e6 for k_, v_ in ks.items() if v_[1]},
                   "sharded": None, "roofline": None, "cpu_baseline": None}
            fp = ks["nifs_fold_prove"]
            if fp[1]:
                ach = (fp[2] / fp[1]) / (fp[0] / fp[1] * 1e-3) / 1e9
                res["roofline"] = {"bound": "hbm", "kernel": "k_nifs_fold_prove (merged fold + prove rounds of NeutronNovaNIFS::prove: 384 B per k and prove pair, SURVEY 8(d))",
                                   "achieved": ach, "peak": 8000.0, "unit": "GB/s", "frac": ach / 8000.0, "traffic": None, "launches": fp[1] / args.steps,
                                   "avg_launch_us": fp[0] / fp[1] * 1e3, "alg_bytes_per_launch": fp[2] / fp[1],
                                   "note": "HIP events on the launching stream inside the timed proves (ctx.kernel_stats); per-round launches of the data-parallel rounds"}
            res["host_serial_ms_note"] = ("~14.8 ms of every step is ONE Keccak-256 sponge absorbing the 256 instances' commitments (8.4 MB, 61.7 K dependent permutations: "
                                          "transcript.absorb(b\"U\", U) for every U, src/neutronnova_zk.rs:553-555) before tau exists; nothing on the device can start "
                                          "before tau (tools: SPARTAN_HOST_LAPS=1). The reference pays the same sponge on its CPU")
            if world == 1 and not args.no_cpu_baseline:
                import ctypes as _ct

                import oracle_lib as ol  # test infrastructure, used here only as the reported CPU baseline (a bounded sample: 4 instances of the same shape)

                cores = ol.lib().orc_set_threads(min(spd.cpu_budget(), 32))
                n_cpu = 4
                oshape = ol.OracleShape(distinct[0])
                okey = _ct.c_void_p(ol.lib().orc_hyrax_setup(b"ck", _ct.c_size_t(CW)))
                Wc = np.stack([host.padded_witness_limbs(dims, distinct[i & 1].witness) for i in range(n_cpu)])
                Xc = np.stack([X2[i & 1] for i in range(n_cpu)])
                rc_ = np.stack([rWs[i] for i in range(n_cpu)])
                cc = np.stack([comms[i] for i in range(n_cpu)])
                t1 = time.perf_counter()
                ol.nifs_prove(oshape, okey, cc, Xc, Wc, rc_, True, ol.Transcript(b"neutronnova_prove"), ol.transcript_round_hook(ol.Transcript(b"vc")))
                secs = time.perf_counter() - t1
                res["cpu_baseline"] = {"value": distinct[0].num_cons * n_cpu / secs, "unit": "constraints/s", "cores": cores, "kind": "port",
                                       "sample": f"NeutronNovaNIFS::prove of {n_cpu} of the {n_total} instances (same 2^{dims['num_cons'].bit_length() - 1}-constraint shape, layers "
                                                 f"built inside the call) on the CPU oracle (C++ restatement, OpenMP over {cores} threads): {secs * 1e3:.0f} ms",
                                       "ms": secs * 1e3, "instances": n_cpu}
            print(json.dumps(res))
        host.nifs_free(prepared)
        comm.close()
        ctx.close()
        group.close()
        return

    if args.workload == "c4":
        # ---- ONE 2^22 proof over all ranks (strong scaling): the timed region is the sharded prove
        inst = _c4_instance()
        sn = host.ShardedSpartanSNARK(ctx, comm, inst)
        tape = np.random.default_rng(0xDEADBEEF).integers(0, 256, size=(8192, 64), dtype=np.uint8)
        used = sn.prep_prove(tape)
        step_tape = np.random.default_rng(0xDEADBEF0).integers(0, 256, size=(8192, 64), dtype=np.uint8)
        for _ in range(args.warmup):
            words, _, _ = sn.prove(step_tape)
        ex0 = comm.stats()["exchanges"]
        barrier()
        t0 = time.perf_counter()
        acc = {}
        for _ in range(args.steps):
            words, _, ph = sn.prove(step_tape)
            for k_, v_ in ph.items():
                acc[k_] = acc.get(k_, 0.0) + v_
        barrier()
        elapsed = group.max_over_ranks(time.perf_counter() - t0)
        legs = None if args.no_sharded else sharded_legs(ctx, comm, group, 1, 2, False)
        unsharded = None
        if world == 1:
            # the SAME instance through the unsharded driver (host/spartan_snark.cpp) next to the sharded one at world 1 - are the
            # synthetic matrices that much heavier than sha256's at the same padded size, or does the sharded driver lack the latency work?
            us = host.SpartanSNARK(ctx, inst)
            assert us.prep_prove(tape) == used
            us.set_flags(prefix_cache=False)
            for _ in range(max(args.warmup, 3)):
                uw, _, _ = us.prove(step_tape)
            uacc, t1 = {}, time.perf_counter()
            for _ in range(args.steps):
                uw, _, uph = us.prove(step_tape)
                for k_, v_ in uph.items():
                    uacc[k_] = uacc.get(k_, 0.0) + v_
            ums = (time.perf_counter() - t1) / args.steps * 1e3
            unsharded = {"driver": "SpartanSNARK (host/spartan_snark.cpp), same instance, same tapes", "ms_per_step": ums,
                         "phases_ms": {k_: v_ / args.steps for k_, v_ in uacc.items()}, "proof_identical_to_the_sharded_one": bool((np.asarray(uw) == np.asarray(words)).all()),
                         "sharded_over_unsharded": (elapsed / args.steps * 1e3) / ums}
            us.close()
        if rank == 0:
            out = {"metric": "synthetic R1CS 2^22 prove(), one proof sharded over the GPUs: R1CS constraints/sec", "value": inst.num_cons * args.steps / elapsed,
                   "unit": "constraints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                   "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                   "dtype": "u256 modular integer (8 x u32 Montgomery limbs; T256 scalar/base fields)",
                   "data": "synthetic: seeded SHA-like R1CS (seed 0xDEADBEEF), Bernoulli(1/2) witness bits, seeded randomness tape",
                   "config": {"workload": "synthetic R1CS 2^22 (BASELINE config 4), SpartanSNARK::prove sharded by row / table slice / column / point range",
                              "num_cons_unpadded": inst.num_cons, "num_cons": sn.dims["num_cons"], "parallelism": f"one proof over {world} ranks, RCCL all-gather per round",
                              "rccl_ranks": world, "exchanges_per_prove": (comm.stats()["exchanges"] - ex0) / args.steps},
                   "phases_ms": {k_: v_ / args.steps for k_, v_ in acc.items() if k_ != "exchanges"}, "unsharded_same_instance": unsharded, "sharded": legs, "roofline": None,
                   "cpu_baseline": None}
            print(json.dumps(out))
        sn.close()
        comm.close()
        ctx.close()
        group.close()
        return

    msg = bytes(args.message_bytes)
    inst = frontend.sha256_circuit(msg)
    t0 = time.time()
    snark = host.SpartanSNARK(ctx, inst)
    t_setup = time.time() - t0
    rng_seed = 0xDEADBEEF + rank
    tape = np.random.default_rng(rng_seed).integers(0, 256, size=(4096, 64), dtype=np.uint8)
    t0 = time.time()
    used = snark.prep_prove(tape)
    t_prep = time.time() - t0
    step_tape = np.random.default_rng(rng_seed + 1).integers(0, 256, size=(4096, 64), dtype=np.uint8)
    snark.set_flags(prefix_cache=False)  # the timed prove() re-hashes the transcript prefix, as the reference's prove does

    # The harness, not the prover: with PyTorch imported, one full collection of CPython's cyclic GC walks a few million objects - 50-70 ms, once
    # every ~100 proves (the ctypes calls allocate), which a mean over 20 steps either misses or takes as +3 ms per step. Collect now, then keep
    # the collector out of the timed loops; step_ms_distribution shows what remains.
    import gc

    gc.collect()
    gc.freeze()
    gc.disable()
    for _ in range(args.warmup):
        words, _, _ = snark.prove(step_tape)
    ctx.reset_stats(True)
    ctx.stats_filter("bind_stream_cubic")  # only the roofline kernel carries HIP events inside the timed region
    barrier()
    t0 = time.perf_counter()
    phase_acc = {}
    step_ms = []  # each prove's own wall time around the call: shows an outlier the mean would hide
    for _ in range(args.steps):
        t_s = time.perf_counter()
        words, _, phases = snark.prove(step_tape)
        step_ms.append((time.perf_counter() - t_s) * 1e3)
        if step_ms[-1] >= max(step_ms):
            slowest = dict(phases)
        for k, v in phases.items():
            phase_acc[k] = phase_acc.get(k, 0.0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = group.max_over_ranks(elapsed)
    bind_ms, bind_launches, bind_bytes = ctx.kernel_stats("bind_stream_cubic")
    ctx.reset_stats(False)
    # the same loop with the transcript prefix's sponge state cached across proves (an API-level optimisation the reference does not make):
    # reported, never the headline
    snark.set_flags(prefix_cache=True)
    snark.prove(step_tape)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        words_c, _, _ = snark.prove(step_tape)
    barrier()
    elapsed_cached = group.max_over_ranks(time.perf_counter() - t0)
    snark.set_flags(prefix_cache=False)
    # The reference-order driver (spartan_snark.cpp prove_reference_order): ONE thread, no helper jobs, no prep-time tables, only include/spartan_hip.h
    # entry points called in the order of the statements of src/spartan.rs:226-466 — what an unchanged spartan.rs bound to the ABI gets. Reported beside
    # the headline in every line; the headline is the C++ driver that arranges more overlap above the ABI.
    snark.set_flags(reference_order=True)
    words_r, _, _ = snark.prove(step_tape)
    snark.prove(step_tape)
    barrier()
    t0 = time.perf_counter()
    ref_phase = {}
    ref_step_ms = []
    for _ in range(args.steps):
        t_s = time.perf_counter()
        words_r, _, ph_r = snark.prove(step_tape)
        ref_step_ms.append((time.perf_counter() - t_s) * 1e3)
        if ref_step_ms[-1] >= max(ref_step_ms):
            ref_slowest = dict(ph_r)
        for k, v in ph_r.items():
            ref_phase[k] = ref_phase.get(k, 0.0) + v
    barrier()
    elapsed_ref = group.max_over_ranks(time.perf_counter() - t0)
    snark.set_flags(reference_order=False)
    gc.enable()
    # untimed extra pass with every kernel class instrumented (main and auxiliary streams), for the per-kernel breakdown
    ctx.reset_stats(True)
    ctx.stats_filter("")
    nb = 3
    for _ in range(nb):
        snark.prove(step_tape)
    kstats = {k: ctx.kernel_stats(k) for k in KERNEL_CLASSES}
    ctx.reset_stats(False)
    # Extra leg, outside the timed region: a single prove is a latency chain (41 host <-> device round trips) that leaves most of the GPU idle,
    # so several independent proofs (one sp_ctx + one host thread each, as the reference would run one rayon pool per proof) overlap well.
    conc = None
    if args.concurrent > 1 and world == 1:
        # In a process of its own (tools/concurrency_stress.py): with PyTorch loaded in the process the same eight contexts measure 10-30 % slower
        # (0.66 vs 0.74-0.81 ms per proof with nothing but `import torch` + a device touch added: profiles/r02_concurrency_notes.txt); the library's
        # callers (the C++ drivers, a Rust host) do not carry PyTorch. Same instance and tapes: the tool's proofs must hash to the timed proof.
        try:
            import hashlib
            import subprocess

            per = max(20, args.steps)
            # every context in flight keeps one owner thread polling its result slots: past the cgroup's CPU quota (16 on the bench boxes) the whole
            # process is throttled and resident kernels run into their 2 s watchdog (measured: 16 contexts -> 39 errors in 320 proofs)
            n_ctx = max(2, min(args.concurrent, spd.cpu_budget() // 2))
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "concurrency_stress.py"), "--contexts", str(n_ctx), "--proofs", str(per),
                                "--message-bytes", str(args.message_bytes), "--tape-seed", str(rng_seed), "--step-seed", str(rng_seed + 1), "--device", str(local_rank),
                                "--json"], capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if not line:
                raise RuntimeError((r.stderr or r.stdout)[-400:])
            cj = json.loads(line[-1])
            same = cj["mismatches"] == 0 and cj["proof_sha256"] == hashlib.sha256(np.ascontiguousarray(words).tobytes()).hexdigest()
            conc = {"proofs_in_flight": cj["proofs_in_flight"], "proofs": cj["proofs"], "constraints_per_s": cj["proofs"] * inst.num_cons / cj["seconds"],
                    "ms_per_proof_amortised": cj["ms_per_proof_amortised"], "proofs_identical_to_the_timed_one": same, "errors": cj["errors"],
                    "process": "tools/concurrency_stress.py (no PyTorch in the process)"}
            # the same leg without one spinning CPU per proof: 24 proofs in flight on 6 polling threads, each proof on a stack of its own and the library's
            # wait hook switching between them (ss_prove_multiplexed). What bounds it is stated with it.
            try:
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "multiplex_bench.py"), "--sweep", "24x6", "--proofs", str(per), "--message-bytes",
                                    str(args.message_bytes), "--device", str(local_rank), "--json"], capture_output=True, text=True, timeout=600,
                                   env=dict(os.environ, GPU_MAX_HW_QUEUES="16"))
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
                if not line:
                    raise RuntimeError((r.stderr or r.stdout)[-400:])
                mj = json.loads(line[-1])["results"][0]
                alg_gb = 1.2  # SURVEY 8(d) bytes of one config-2 prove (outer 403 MB + inner 268 MB + incremental SpMV 210 MB + poly_ABC 248 MB + eq / rowmat)
                conc["multiplexed"] = {"proofs_in_flight": mj["proofs_in_flight"], "polling_threads": mj["polling_threads"], "proofs": mj["proofs"],
                                       "ms_per_proof_amortised": mj["ms_per_proof_amortised"], "constraints_per_s": inst.num_cons / (mj["ms_per_proof_amortised"] * 1e-3),
                                       "mismatches": mj["mismatches"], "rc": mj["rc"],
                                       "aggregate_hbm_roofline_frac": (alg_gb / (mj["ms_per_proof_amortised"] * 1e-3)) / 8000.0 if args.message_bytes == 2048 else None,
                                       "bound": "GPU time: the kernels of one prove sum to ~0.55 ms when each runs alone (profiles/r04_mux_kernel_stats.md); more proofs in "
                                                "flight, more polling threads, a larger resident-tail budget or a second process do not move it"}
            except Exception as exc:
                conc["multiplexed"] = {"error": repr(exc)}
        except Exception as exc:  # an extra must never cost the bench line
            conc = {"error": repr(exc)}

    # roofline of the same kernel on tables past the 256 MiB Infinity Cache: prove_cubic_with_three_inputs on three 2^23-entry tables (768 MiB);
    # its first fused launch (bind round 1 + evaluate round 2 over 2^23-entry tables) is accounted as "bind_stream_cubic_hbm"
    hbm = None
    try:
        rr = np.random.default_rng(5)
        mk = lambda: hip.Table.eq(ctx, rr.integers(0, 1 << 62, size=(23, 4), dtype=np.uint64))
        ctx.reset_stats(True)
        ctx.stats_filter("bind_stream_cubic_hbm")
        reps = 3
        for _ in range(reps):
            A, B, C = mk(), mk(), mk()
            hip.sumcheck_cubic3(ctx, np.zeros(4, dtype=np.uint64), rr.integers(0, 1 << 62, size=(23, 4), dtype=np.uint64), A, B, C, hip.Transcript(ctx, b"roofline"))
            for t_ in (A, B, C):
                t_.free()
        hms, hl, hb = ctx.kernel_stats("bind_stream_cubic_hbm")
        ctx.reset_stats(False)
        ctx.stats_filter("")
        if hl:
            ach = (hb / hl) / (hms / hl * 1e-3) / 1e9
            hbm = {"bound": "hbm", "kernel": "k_bind_eval_cubic_stream on three 2^23-entry tables (768 MiB: past the 256 MiB Infinity Cache)", "achieved": ach, "peak": 8000.0,
                   "unit": "GB/s", "frac": ach / 8000.0, "launches": hl, "avg_launch_us": hms / hl * 1e3, "alg_bytes_per_launch": hb / hl}
    except Exception as exc:  # an extra must never cost the bench line
        hbm = {"error": repr(exc)}
        ctx.reset_stats(False)
        ctx.stats_filter("")
    # SpartanSNARK::verify on the device-backed path (reported separately, as the reference's bench does: benches/sha256_spartan.rs:245-262)
    v_ok = snark.verify(words) == 0  # warm-up: first use of the verifier's workspaces
    t0 = time.perf_counter()
    v_ok = v_ok and all(snark.verify(words) == 0 for _ in range(3))
    t_verify = (time.perf_counter() - t0) / 3

    out = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        ncons = inst.num_cons
        value = spd.whole_job_throughput(ncons, args.steps, elapsed, world)
        achieved = (bind_bytes / bind_launches) / (bind_ms / bind_launches * 1e-3) / 1e9 if bind_launches else 0.0
        # HBM bytes per launch of the roofline kernel from the PMC passes committed under profiles/ (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
        # in separate runs, corrected as MI355X_MICROARCH.md prescribes: 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024); null if absent
        traffic, traffic_src = None, None
        for name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
            pmc = os.path.join(ROOT, "profiles", name)
            if os.path.exists(pmc) and args.message_bytes == 2048:
                with open(pmc) as f:
                    tj = json.load(f)
                keys = [k for k in tj if "k_bind_eval_cubic_stream<1" in k and k.endswith("@262144")]
                if keys:
                    traffic, traffic_src = tj[keys[0]]["traffic_bytes"], f"profiles/{name} (separate rocprofv3 --pmc passes of this command)"
                    break
        # rocprofv3 kernel-only durations of the same call sites, committed under profiles/ (tools/profile_job.sh): inside a prove and with the GPU
        # otherwise empty. The HIP-event figures below are taken inside a prove, where the auxiliary streams' kernels are resident beside these.
        rocprof_sites = None
        sites_path = next((p_ for p_ in (os.path.join(ROOT, "profiles", n_) for n_ in ("r06_kernel_sites.json", "r05_kernel_sites.json", "r04_kernel_sites.json")) if os.path.exists(p_)), "")
        if os.path.exists(sites_path) and args.message_bytes == 2048:
            with open(sites_path) as f:
                sj = json.load(f)
            pick = {"poly_abc": "k_polyabc_short_and_long", "rowmat_vec": "k_rowmat_vec_tall", "eval_quad": "k_eval_quad_stream_lowhi<4>", "eval_cubic": "k_eval_products_stream<1",
                    "bind_stream_quad": "k_bind_eval_quad_stream@", "bind_stream_quad_sparse": "k_bind_eval_quad_stream_sparse", "spmv_incremental": "k_spmv3"}
            rocprof_sites = {}
            for cls, sub in pick.items():
                ks = [k for k in sj if k.startswith(sub)]
                if ks:
                    v = sj[ks[0]]
                    rocprof_sites[cls] = {"site": ks[0], "in_prove_us": v["in_prove_median_us"], "alone_us": v["solo_median_us"], "others_resident_frac": v["others_resident_frac"],
                                          "alone_GBps": (v["alg_bytes"] / v["solo_median_us"] / 1e3) if v["alg_bytes"] and v["solo_median_us"] else None}
        si = snark.shape_info
        gbps = lambda k: (kstats[k][2] / max(kstats[k][0], 1e-9)) / 1e6  # algorithmic bytes / event-timed ms -> GB/s
        out = {
            "metric": "sha256_spartan prove(): R1CS constraints/sec (prove wall-clock in ms_per_step)",
            "value": value,
            "unit": "constraints/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u256 modular integer (8 x u32 Montgomery limbs; T256 scalar/base fields)",
            "data": "synthetic: all-zero message of the bench (benches/sha256_spartan.rs:172), own SHA-256 R1CS generator, seeded randomness tape",
            "config": {"workload": f"sha256_spartan {args.message_bytes} B, SpartanSNARK::prove on T256HyraxEngine shapes", "num_cons_unpadded": ncons,
                       "num_cons": snark.dims["num_cons"], "num_vars": snark.dims["num_shared"] + snark.dims["num_precommitted"] + snark.dims["num_rest"],
                       "parallelism": f"{world} independent proofs (one per GPU)", "host_cpus_local_to_gpu": numa_cpus,
                       "transcript_prefix": "re-hashed in every timed prove (reference-equivalent)"},
            "transcript_prefix_cached": {"ms_per_step": elapsed_cached / args.steps * 1e3, "constraints_per_s": world * ncons * args.steps / elapsed_cached,
                                         "proof_identical": bool((words_c == words).all()),
                                         "note": "sponge state of new + vk + public_values + comm_W_precommitted kept across proves (FLAG_PREFIX_CACHE): not the headline"},
            "step_ms_distribution": dict(dist(step_ms), slowest_step_phases_ms=slowest),
            "reference_order": {"ms_per_step": elapsed_ref / args.steps * 1e3, "constraints_per_s": world * ncons * args.steps / elapsed_ref,
                                "step_ms_distribution": dict(dist(ref_step_ms), slowest_step_phases_ms=ref_slowest),
                                "proof_identical": bool((words_r == words).all()), "phases_ms": {k: v / args.steps for k, v in ref_phase.items()},
                                "headline_over_reference_order": elapsed / elapsed_ref,
                                "note": "one thread, statement order of src/spartan.rs:226-466, ABI calls only (PCS::prove = one sp_hyrax_prove): the time of an unchanged spartan.rs over the ABI"},
            "roofline": {"bound": "hbm", "kernel": "k_bind_eval_cubic_stream<1> (outer sum-check: bind round 1 fused with the evaluation of round 2, 3 tables of 2^20)",
                         "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": traffic,
                         "traffic_source": traffic_src, "launches": bind_launches, "avg_launch_us": bind_ms / max(bind_launches, 1) * 1e3,
                         "alg_bytes_per_launch": bind_bytes / max(bind_launches, 1),
                         "other_kernels": {k: {"launches_per_step": kstats[k][1] / nb, "avg_us": kstats[k][0] / max(kstats[k][1], 1) * 1e3, "alg_GBps": gbps(k)}
                                           for k in ("bind_stream_quad", "bind_stream_quad_sparse", "bind", "eval_cubic", "eval_quad", "spmv_incremental", "poly_abc",
                                                     "rowmat_vec") if kstats[k][1]},
                         "other_kernels_rocprof": rocprof_sites,
                         "other_kernels_note": "avg_us are HIP-event times of an instrumented pass INSIDE a prove: kernels of the auxiliary streams (delta's MSM, the PCS table "
                                               "walks, the resident sum-check tail) are resident beside them, so they are upper bounds of the kernel's own time; "
                                               "other_kernels_rocprof (profiles/r06_kernel_stats.md) gives the rocprofv3 kernel-only duration of the same call site inside a prove "
                                               "and alone. The 'bind' class (fused bind + evaluate launches on tables <= 2^19 elements) is launched AHEAD of "
                                               "its challenge and waits for it at the mailbox: its avg_us includes that wait, so its GB/s understate the kernel; the streaming "
                                               "classes and the roofline kernel (tables >= 2^20) are launched behind their challenge and their times are the kernels' alone."},
            "roofline_hbm": hbm,
            "sparse": {"nnz_A_B_C": si["nnz"], "nnz_filtered_A_B_C": si["nnz_filtered"], "long_columns": si["long_columns"],
                       "spmv_incremental_alg_bytes": kstats["spmv_incremental"][2] / max(kstats["spmv_incremental"][1], 1),
                       "poly_abc_alg_bytes": kstats["poly_abc"][2] / max(kstats["poly_abc"][1], 1),
                       "accounting": "SURVEY 8(d): 36 B per entry (index + gathered element) + 32 B per output (+ 32 B per cached input)"},
            "phases_ms": {k: v / args.steps for k, v in phase_acc.items()},
            "kernel_ms_per_step": {k: v[0] / nb for k, v in kstats.items()},
            "setup_s": t_setup,
            "prep_prove_s": t_prep,
            "concurrent_proofs_extra": conc,
            "sharded": None,
            "verify_ms": t_verify * 1e3,
            "verify_accepts": v_ok,
        }
        if world == 1 and not args.no_cpu_baseline:
            import oracle_lib as ol  # test infrastructure, used here only as the reported CPU baseline

            osp = ol.OracleSpartan(inst)
            ou = osp.prep_prove(tape)
            assert ou == used
            # the oracle's loops are OpenMP-parallel where the reference's are rayon-parallel; exact arithmetic, so the proof does not depend
            # on the thread count. Timed: second prove (first one pays the page faults) at all host cores, then one at a single thread.
            cores = ol.lib().orc_set_threads(min(spd.cpu_budget(), 32))  # OpenMP threads past the cgroup's CPU quota only get the whole process throttled
            osp.prove(step_tape)
            want, _, secs = osp.prove(step_tape)
            ol.lib().orc_set_threads(1)
            _, _, secs1 = osp.prove(step_tape)
            ol.lib().orc_set_threads(cores)
            ok = bool((want == words).all()) and osp.verify_words(words) == 0
            out["cpu_baseline"] = {"value": ncons / secs, "unit": "constraints/s", "cores": cores, "kind": "port",
                                   "sample": f"full prove() of the same {args.message_bytes} B instance on the CPU oracle (C++ restatement, OpenMP over {cores} threads: "
                                             f"{secs * 1e3:.0f} ms; single thread: {secs1 * 1e3:.0f} ms); 3 proves + prep_prove of CPU work in all",
                                   "ms": secs * 1e3, "single_thread_ms": secs1 * 1e3, "gpu_proof_bit_exact_and_verified": ok}
            if not ok:
                raise SystemExit("GPU proof differs from the oracle's or fails verification")
            out["prep_prove"] = prep_prove_leg(snark, inst, tape, step_tape, words, oracle=osp, used=used)
            out["prep_prove"]["cpu_oracle_threads"] = cores
        elif world == 1:
            out["prep_prove"] = prep_prove_leg(snark, inst, tape, step_tape, words, used=used)
    if not args.no_sharded:
        # The sharded legs are the one part of this run that talks RCCL from C++ across ranks. They come after everything the headline needs, each
        # under a watchdog of its own (LegDog): a leg that never returns costs only itself - rank 0 prints the line with the legs finished before it
        # (the commit leg, the north-star's MSM figure, runs first) and every rank leaves.
        legs = {}
        if rank == 0:
            out["sharded"] = legs
        guardian = LineGuardian() if rank == 0 else None
        dog = LegDog(rank, out if rank == 0 else None, legs, args.extras_timeout / 2, guardian)
        try:
            dog.arm("communicator")
            comm = host.Comm(rank, world, comm_backend, device=local_rank)
            sharded_legs(ctx, comm, group, 3, 2, world == 1 and not args.no_cpu_baseline, out=legs, dog=dog)
        except Exception as exc:
            legs["error"] = repr(exc)
        dog.disarm()
        if guardian:
            guardian.done()
    if rank == 0:
        out["scaling_vs_1"] = None if rehearsal else _scaling_vs_1(world, args, out)
        if rehearsal:
            out["rehearsal"] = "SPARTAN_BENCH_REHEARSAL=1: all ranks on ONE GPU over gloo + the callback exchange; control flow only, the figures are not measurements"
        print(json.dumps(out))
    snark.close()
    if comm is not None:
        comm.close()
    ctx.close()
    group.close()


if __name__ == "__main__":
    main()
