"""The expensive ORACLE halves of the GPU parity tests as jobs that can run beside the tests.

The oracle (oracle/srukf_oracle.c: single thread, the reference's formulation) needs 10 - 65 s per case at N >= 300; by round 5 the `-m gpu` suite spent 600 of its
770 s waiting for it, one case after the other, on a box with 16 cores.  A case's oracle half depends on nothing the device computes, so the session starts all of
them at once in child processes (OraclePool: `python tests/oracle_jobs.py <kind> <json kwargs> <out.npz>`, a few at a time) and a test picks its result up when it
gets there.  Same functions, same inputs, same numbers as the inline calls they replace; a case that was not started ahead (a single test selected with -k) runs
inline.  Test infrastructure: nothing here is used by the product path."""
import json
import os
import subprocess
import sys
import tempfile
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _mods():
    sys.path.insert(0, ROOT)
    import __graft_entry__ as ge
    from oracle import oracle as O
    O.lib()
    return ge.load_package().synth, O


def two_frames(N, storage="f64", F=2, seed=3):
    """oracle half of test_gpu_parity_r5._run_two_frames_against_oracle: F BATCHED frames with a quarter of the landmarks unmatched per frame; fp32 storage: the
    state that lives from frame to frame is rounded to float."""
    synth, O = _mods()
    p = synth.scene_params()
    sc = synth.make_scene(N, F, seed=seed, p=p)
    rng = np.random.default_rng(1000 + N)
    matched = np.ones((F, N), dtype=np.int32)
    for t in range(F):
        matched[t, rng.permutation(N)[:N // 4]] = 0
    X0, S0 = sc["X0"], np.triu(sc["S0"])
    if storage == "f32":
        X0, S0 = X0.astype(np.float32).astype(np.float64), S0.astype(np.float32).astype(np.float64)
    o = O.Oracle(N, p); o.set_state(X0, S0)
    n = 6 * N + 4
    to = np.zeros((F, 8))
    for t in range(F):
        tr = o.run_frames(sc["odo"][t:t + 2], sc["z"][t:t + 1], matched[t:t + 1], O.Oracle.BATCHED)
        to[t] = tr[0]
        if storage == "f32":
            Xo, So = o.get_state()
            Xo, So = Xo.astype(np.float32).astype(np.float64), np.triu(So).astype(np.float32).astype(np.float64)
            o.set_state(Xo, So)
            to[t, :4] = Xo[n - 4:]; to[t, 4:] = (So[:, n - 4:n - 2].T @ So[:, n - 4:n - 2]).ravel()
    Xo, So = o.get_state()
    return {"to": to, "Xo": Xo, "So": So}


def one_frame(N, seed=0, storage="f64", eps=None, mode=1):
    """one whole frame from the scene's initial state (float-rounded for fp32 storage): h / Si / visible after the predict half, X and S after the update
    (test_gpu_parity_r2.test_oracle_frame_n500; the mixed-precision tests)."""
    synth, O = _mods()
    p = synth.scene_params()
    if eps is not None:
        p["epsilon"] = eps
    sc = synth.make_scene(N, 1, seed=seed, p=p)
    X0, S0 = sc["X0"], sc["S0"]
    if storage == "f32":
        X0, S0 = X0.astype(np.float32).astype(np.float64), np.triu(S0).astype(np.float32).astype(np.float64)
    o = O.Oracle(N, p); o.set_state(X0, S0)
    o.predict_motion(sc["odo"][0], sc["odo"][1])
    ho, Sio, viso = o.predict_measurement()
    o.update(sc["z"][0], sc["matched"][0], 1, 0, mode)
    Xo, So = o.get_state()
    return {"h": ho, "Si": Sio, "vis": viso, "Xo": Xo, "So": So}


def batched_filter(b, N=200, F=12, Fo=2):
    """oracle half of test_gpu_parity_r4.test_batched_filters_against_the_oracle_and_against_solo_runs for filter b: the first Fo BATCHED frames of its sequence"""
    synth, O = _mods()
    p = synth.scene_params()
    sc = synth.make_scene(N, F, seed=0, p=p, obs_seed=7000 + b)
    o = O.Oracle(N, p); o.set_state(sc["X0"], sc["S0"])
    to = o.run_frames(sc["odo"][:Fo + 1], sc["z"][:Fo], sc["matched"][:Fo], O.Oracle.BATCHED)
    Xo, So = o.get_state()
    return {"to": to, "Xo": Xo, "So": So}


KINDS = {"two_frames": two_frames, "one_frame": one_frame, "batched_filter": batched_filter}


def _key(kind, kw):
    return kind + ":" + json.dumps(kw, sort_keys=True)


class OraclePool:
    """starts jobs in child processes (at most `workers` at a time, in submission order) and hands their results out"""

    def __init__(self, workers=None):
        ncpu = os.cpu_count() or 4
        try:
            q, per = open("/sys/fs/cgroup/cpu.max").read().split()
            if q != "max":
                ncpu = min(ncpu, int(float(q) / float(per)))
        except Exception:
            pass
        self.workers = workers or max(2, min(12, ncpu - 3))
        self.tmp = tempfile.mkdtemp(prefix="oracle_jobs_")
        self.jobs, self.order, self.lock = {}, [], threading.Lock()
        self.sem = threading.Semaphore(self.workers)
        self.closed = False

    def submit(self, kind, **kw):
        k = _key(kind, kw)
        with self.lock:
            if k in self.jobs or self.closed:
                return
            out = os.path.join(self.tmp, f"job{len(self.jobs)}.npz")
            job = {"out": out, "done": threading.Event(), "rc": None}
            self.jobs[k] = job
        th = threading.Thread(target=self._run, args=(kind, kw, job), daemon=True)
        job["thread"] = th
        th.start()

    def _run(self, kind, kw, job):
        with self.sem:
            if self.closed:
                job["rc"] = -1; job["done"].set(); return
            env = dict(os.environ, OMP_NUM_THREADS="1")
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), kind, json.dumps(kw), job["out"]], env=env, capture_output=True, text=True, timeout=1500)
                job["rc"], job["err"] = r.returncode, r.stderr[-400:]
            except Exception as e:                                 # noqa: BLE001
                job["rc"], job["err"] = -2, repr(e)
        job["done"].set()

    def get(self, kind, **kw):
        job = self.jobs.get(_key(kind, kw))
        if job is not None:
            job["done"].wait()
            if job["rc"] == 0:
                with np.load(job["out"]) as z:
                    return {k: z[k] for k in z.files}
        return KINDS[kind](**kw)                                   # not started ahead (or the child failed): inline, as before

    def close(self):
        self.closed = True
        import shutil
        shutil.rmtree(self.tmp, ignore_errors=True)


if __name__ == "__main__":
    kind, kw, out = sys.argv[1], json.loads(sys.argv[2]), sys.argv[3]
    res = KINDS[kind](**kw)
    np.savez(out + ".tmp.npz", **res)
    os.replace(out + ".tmp.npz", out)
