"""CLI invocations of the GPU suite: as child processes (`run_parallel`) or in this process (`call_main`).

`run_parallel` runs its jobs ONE AT A TIME by default: four driver processes side by side on the box's one GPU were measured
SLOWER than one after the other (98 s vs 74 s for the eight runs of the SDXL driver-equivalence test: the processes time-slice
the GPU and each spends its seconds in graph capture and launches, not in Python).  Every job gets its OWN working directory
when it writes to `./exp`.

`call_main` runs a script's `main(argv)` inside the pytest process (no interpreter start, torch import or HIP initialisation:
2-3 s instead of 9-12 s per run).  Every script also keeps at least one test that starts it as `python <script>`; the
equivalence tests (batched / in-flight schedules == the per-image loop), which need several runs each, use `call_main`.
"""
import contextlib
import importlib.util
import io
import os
import subprocess
import sys
import time


class Done:
    def __init__(self, args, returncode, stdout, stderr):
        self.args, self.returncode, self.stdout, self.stderr = args, returncode, stdout, stderr

    def last_json(self):
        import json
        return json.loads(self.stdout.strip().splitlines()[-1])


def run_parallel(jobs, max_procs=1, timeout=900, env=None):
    """jobs: list of (argv after the interpreter, cwd).  Returns the list of `Done` in job order; asserts every return code."""
    e = dict(os.environ)
    e.update(env or {})
    pending = list(enumerate(jobs))
    running, done = [], [None] * len(jobs)
    t0 = time.time()
    try:
        while pending or running:
            while pending and len(running) < max_procs:
                i, (args, cwd) = pending.pop(0)
                os.makedirs(str(cwd), exist_ok=True)
                fo, fe = open(os.path.join(str(cwd), f"job{i}.out"), "w+"), open(os.path.join(str(cwd), f"job{i}.err"), "w+")
                p = subprocess.Popen([sys.executable] + list(args), cwd=str(cwd), stdout=fo, stderr=fe, text=True, env=e)
                running.append((i, args, p, fo, fe))
            still = []
            for i, args, p, fo, fe in running:
                rc = p.poll()
                if rc is None:
                    still.append((i, args, p, fo, fe))
                    continue
                fo.seek(0), fe.seek(0)
                done[i] = Done(args, rc, fo.read(), fe.read())
                fo.close(), fe.close()
            running = still
            if running:
                if time.time() - t0 > timeout:
                    raise subprocess.TimeoutExpired(running[0][1], timeout)
                time.sleep(0.05)
    finally:
        for _, _, p, fo, fe in running:
            p.kill()
            p.wait()
    for d in done:
        assert d.returncode == 0, (d.args, d.stdout[-2000:], d.stderr[-4000:])
    return done


_BARE = ("_bootstrap", "edit_real", "edit_syn", "test", "sd_mapping")       # modules the scripts import by bare name from their own folder


def call_main(script, argv, cwd):
    """`python <script> <argv>` with working directory `cwd`, but inside this process: the script is loaded under a private
    module name with its folder (and p2p/, where `_bootstrap` lives) in front of sys.path, `main(argv)` runs with stdout
    captured.  Returns a `Done` (returncode 0; exceptions propagate)."""
    script = os.path.abspath(script)
    folder = os.path.dirname(script)
    p2p = os.path.join(os.path.dirname(folder), "p2p")
    saved_mods = {k: sys.modules.pop(k) for k in _BARE if k in sys.modules}
    saved_path, saved_cwd = list(sys.path), os.getcwd()
    os.makedirs(str(cwd), exist_ok=True)
    out = io.StringIO()
    try:
        sys.path[:0] = [folder, p2p]
        os.chdir(str(cwd))
        name = "_cli_" + os.path.basename(folder) + "_" + os.path.splitext(os.path.basename(script))[0]
        spec = importlib.util.spec_from_file_location(name, script)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        with contextlib.redirect_stdout(out):
            spec.loader.exec_module(mod)
            mod.main([str(a) for a in argv])
    finally:
        os.chdir(saved_cwd)
        sys.path[:] = saved_path
        for k in _BARE:
            sys.modules.pop(k, None)
        sys.modules.update(saved_mods)
    return Done([script] + list(argv), 0, out.getvalue(), "")


def run_mixed(jobs, in_process=(), **kw):
    """`run_parallel(jobs)` with the jobs whose index is in `in_process` run by `call_main` instead (the FIRST run of a script
    in a test stays a real `python <script>` child; repeats with other flags need not pay the interpreter start again)"""
    out = []
    for i, (args, cwd) in enumerate(jobs):
        out.append(call_main(args[0], args[1:], cwd) if i in in_process else run_parallel([(args, cwd)], **kw)[0])
    return out
