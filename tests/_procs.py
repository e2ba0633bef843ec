"""Run independent CLI invocations of the GPU suite side by side.

A driver process spends most of its seconds importing torch, initialising HIP and capturing its step graph, not on the GPU;
the suite's CLI tests are independent invocations, so they run `max_procs` at a time (the GPU box allows 6 processes on the
card: pytest itself + 4 children stays inside that).  Every job gets its OWN working directory when it writes to `./exp`.
"""
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


def run_parallel(jobs, max_procs=4, timeout=900, env=None):
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
