"""Process pool for the seed-bank SpectralClustering (host-side, scikit-learn, as in the reference).

The reference fits one SpectralClustering per class and domain inside ``GModule.update_seed``
(models/graph_matching.py:532-567) on the training thread; on an MI355X that host work (~6 ms per fit, up to
2*num_classes fits per step) is longer than the GPU work it blocks.  The fits are independent of each other and their
result is only consumed by the next read of the seed bank, so GModule submits them here and keeps launching kernels;
worker processes (plain ``python _cluster_worker.py``: no torch, no GPU context) run them concurrently.
"""
import atexit
import os
import pickle
import struct
import subprocess
import sys

_WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_cluster_worker.py")


class _Worker:
    def __init__(self):
        env = dict(os.environ)
        env.setdefault("OMP_NUM_THREADS", "2")
        env.setdefault("OPENBLAS_NUM_THREADS", "2")
        self.proc = subprocess.Popen([sys.executable, "-u", _WORKER], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                     env=env)
        self.done = {}

    def send(self, job):
        body = pickle.dumps(job, protocol=pickle.HIGHEST_PROTOCOL)
        self.proc.stdin.write(struct.pack("<Q", len(body)) + body)
        self.proc.stdin.flush()

    def _read(self, n):
        buf = b""
        while len(buf) < n:
            chunk = self.proc.stdout.read(n - len(buf))
            if not chunk:
                raise RuntimeError("cluster worker exited")
            buf += chunk
        return buf

    def wait(self, job_id):
        while job_id not in self.done:
            n = struct.unpack("<Q", self._read(8))[0]
            jid, keep, err = pickle.loads(self._read(n))
            self.done[jid] = (keep, err)
        return self.done.pop(job_id)

    def close(self):
        try:
            self.proc.stdin.close()
            self.proc.terminate()
        except Exception:
            pass


class ClusterPool:
    """submit(rows, n_neighbors) -> ticket; result(ticket) -> keep mask (falls back to an inline fit on any failure)."""

    def __init__(self, workers=None):
        # 8 = one worker per fit of a GModule call with four classes and two domains: the fits of one call run side by side
        # (config 5 in its stated dtype, 30-step averages on one box: 47.4 / 48.1 ms per step with 4 workers, 41.1 / 42.1 with 8,
        # 45.3 with 16 -- tools/seed_wait.py)
        # Under torchrun every rank has a pool of its own: 8 only where the node has the cores for it (24 per rank), else 4.
        ranks = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
        default = 8 if (os.cpu_count() or 8) // ranks >= 24 else 4
        self.n = int(os.environ.get("GE_CLUSTER_WORKERS", str(default))) if workers is None else int(workers)
        self.workers = []
        self.next_id = 0
        self.inflight = {}
        atexit.register(self.close)

    def submit(self, rows, n_neighbors):
        jid = self.next_id
        self.next_id += 1
        try:
            if len(self.workers) < self.n:
                self.workers.append(_Worker())
            w = self.workers[jid % len(self.workers)]
            w.send((jid, rows, int(n_neighbors)))
            self.inflight[jid] = (w, rows, int(n_neighbors))
        except Exception:
            self.inflight[jid] = (None, rows, int(n_neighbors))
        return jid

    def result(self, jid):
        w, rows, nn = self.inflight.pop(jid)
        if w is not None:
            try:
                keep, err = w.wait(jid)
                if err is None:
                    return keep
            except Exception:
                if w in self.workers:
                    self.workers.remove(w)
                w.close()
        from ._cluster_worker import spectral_keep   # same function, run here

        return spectral_keep(rows, nn)

    def close(self):
        for w in self.workers:
            w.close()
        self.workers = []


_POOL = None


def get_pool():
    global _POOL
    if _POOL is None:
        _POOL = ClusterPool()
    return _POOL
