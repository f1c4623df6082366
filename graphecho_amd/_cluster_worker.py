"""Seed-bank clustering worker: a standalone script (no package imports, no torch) run by cluster_pool.ClusterPool.

Protocol on stdin/stdout: 8-byte little-endian length + pickle.  Request ``(job_id, rows float32 [n+1, d], n_neighbors)``
-> response ``(job_id, keep bool [n])`` where ``keep`` marks the rows that fall in the same spectral cluster as row 0
(the current seed), computed exactly as the reference does (models/graph_matching.py:553-560 of the reference:
``SpectralClustering(2, affinity='nearest_neighbors', assign_labels='kmeans', random_state=1234, n_neighbors=n//2)``; the
reference also passes ``n_jobs=-1``, which only parallelises the neighbour search -- same labels (checked for 60 / 120 / 250 rows),
but inside a two-thread worker joblib's pool start-up showed as 100 ms outliers: left at None here).
A reader thread drains stdin so the parent never blocks on a full pipe while a fit is running.
"""
import pickle
import queue
import struct
import sys
import threading


def spectral_keep(rows, n_neighbors):
    import sklearn.cluster as cluster

    sp = cluster.SpectralClustering(2, affinity="nearest_neighbors", n_jobs=None, assign_labels="kmeans",
                                    random_state=1234, n_neighbors=n_neighbors)
    indx = sp.fit_predict(rows)
    return (indx == indx[0])[1:]


def _read_exact(f, n):
    buf = b""
    while len(buf) < n:
        chunk = f.read(n - len(buf))
        if not chunk:
            return None
        buf += chunk
    return buf


def main():
    fin, fout = sys.stdin.buffer, sys.stdout.buffer
    jobs = queue.Queue()

    def reader():
        while True:
            head = _read_exact(fin, 8)
            if head is None:
                jobs.put(None)
                return
            body = _read_exact(fin, struct.unpack("<Q", head)[0])
            if body is None:
                jobs.put(None)
                return
            jobs.put(pickle.loads(body))

    threading.Thread(target=reader, daemon=True).start()
    import sklearn.cluster  # noqa: F401  (pay the import while the first job is still in flight)

    while True:
        job = jobs.get()
        if job is None:
            return
        job_id, rows, n_neighbors = job
        try:
            out = (job_id, spectral_keep(rows, n_neighbors), None)
        except Exception as exc:  # reported to the parent, which then runs the fit inline
            out = (job_id, None, repr(exc))
        body = pickle.dumps(out, protocol=pickle.HIGHEST_PROTOCOL)
        fout.write(struct.pack("<Q", len(body)) + body)
        fout.flush()


if __name__ == "__main__":
    main()
