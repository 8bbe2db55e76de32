"""CPU tests of the JSON wire format and of the client library's recall / MRR bookkeeping, through the compiled binary
tests/cpp/test_wire (built by make -C prefhetch_amd/csrc).  The recall figures are checked against a numpy
restatement of /root/reference/src/client/client_lib.cpp:243-337."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

BIN = os.path.join(ROOT, "tests", "cpp", "test_wire")
NQUERY, K = 5, 100


def _need_bin():
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "prefhetch_amd", "csrc")])
    assert os.path.exists(BIN)


def test_json_reader_writer_and_error_behaviour():
    _need_bin()
    r = subprocess.run([BIN, "selftest"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "selftest: ok" in r.stdout, r.stdout + r.stderr


def recall_reference(observed, gt, gt_nn):
    """the reference's counting: ground-truth neighbour j < K is a hit at the position k < K where it shows up"""
    rec = np.zeros(3)
    mrr = np.zeros(3, dtype=np.float32)
    for i in range(NQUERY):
        for j in range(K):
            hits = np.nonzero(observed[i] == gt[i * gt_nn + j])[0]
            if len(hits) == 0:
                continue
            k = int(hits[0])
            for a, lim in enumerate((1, 10, 100)):
                if k < lim:
                    rec[a] += 1
                    if j == 0:
                        mrr[a] = np.float32(mrr[a] + np.float32(1.0) / np.float32(k + 1))
    return [np.float32(rec[0] / (1 * NQUERY)), np.float32(rec[1] / (10 * NQUERY)), np.float32(rec[2] / (100 * NQUERY))] + \
           [np.float32(m / np.float32(NQUERY)) for m in mrr]


@pytest.mark.parametrize("seed,gt_nn,overlap", [(1, 100, 1.0), (2, 100, 0.5), (3, 128, 0.9), (4, 100, 0.0), (5, 200, 0.7)])
def test_recall_and_mrr_match_the_reference_definition(tmp_path, seed, gt_nn, overlap):
    _need_bin()
    rng = np.random.default_rng(seed)
    gt = np.stack([rng.permutation(100000)[:gt_nn] for _ in range(NQUERY)]).astype(np.int32)
    observed = np.empty((NQUERY, K), dtype=np.int64)
    for i in range(NQUERY):
        row = gt[i, :K].astype(np.int64).copy()
        rng.shuffle(row[: max(2, int(K * 0.3))])                       # perturb the head
        miss = rng.random(K) >= overlap
        row[miss] = 200000 + rng.integers(0, 1000, miss.sum())          # ids outside the ground truth (may repeat)
        observed[i] = row
    path = tmp_path / "case.bin"
    with open(path, "wb") as f:
        f.write(observed.tobytes())
        f.write(np.int32(gt_nn).tobytes())
        f.write(gt.tobytes())
    r = subprocess.run([BIN, "recall", str(path)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    got = [np.float32(x) for x in r.stdout.split()]
    exp = recall_reference(observed, gt.reshape(-1), gt_nn)
    assert got == exp, (got, exp)


def test_recall_rejects_short_ground_truth(tmp_path):
    _need_bin()
    path = tmp_path / "short.bin"
    with open(path, "wb") as f:
        f.write(np.zeros((NQUERY, K), dtype=np.int64).tobytes())
        f.write(np.int32(50).tobytes())
        f.write(np.zeros((NQUERY, 50), dtype=np.int32).tobytes())
    r = subprocess.run([BIN, "recall", str(path)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 3 and "K greater than nearest neigbours" in r.stdout


def test_http_transport_loopback_cpp_client():
    r = subprocess.run([BIN, "http-loopback"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "http loopback: ok" in r.stdout, r.stdout + r.stderr


def test_http_listener_serves_a_stock_http_client():
    """the POSIX-socket listener (include/server/http.h) against Python's http.client, the way libcurl / cpr talks to the
    reference's Drogon server: GET /query, POST bodies of the reference's shape (Query.cc:29-63), keep-alive, Expect:
    100-continue, unknown route -> 404, handler exception -> 500, bad method -> 405"""
    import http.client
    import json
    n = 9
    p = subprocess.Popen([BIN, "http", str(n)], stdout=subprocess.PIPE, text=True)
    try:
        port = int(p.stdout.readline().split()[1])
        c = http.client.HTTPConnection("127.0.0.1", port, timeout=30)
        c.request("GET", "/query")
        r = c.getresponse()
        assert r.status == 200 and r.getheader("Content-Type") == "application/json" and json.loads(r.read()) == [[1.5, 2.0]]
        body = json.dumps({"preciseQuery": [[float(i % 7) for i in range(128)] for _ in range(5)],
                           "nearestCentroidIndexes": [[i for i in range(20)] for _ in range(5)]})
        c.request("POST", "/echo", body=body, headers={"Content-Type": "application/json"})       # same connection: keep-alive
        r = c.getresponse()
        assert r.status == 200 and r.read().decode() == body
        big = "x" * (2 << 20)
        c.request("POST", "/echo", body=big, headers={"Expect": "100-continue"})                   # what libcurl sends above 1 KB
        r = c.getresponse()
        assert r.status == 200 and r.read().decode() == big
        for route, status in (("/nowhere", 404), ("/boom", 500), ("/badbody", 500)):
            c.request("POST", route, body="{}")
            r = c.getresponse()
            assert r.status == status, (route, r.status)
            r.read()
        c.request("DELETE", "/query")
        r = c.getresponse()
        assert r.status == 405
        r.read()
        c.request("GET", "/query?x=1")                                                             # query strings are ignored
        r = c.getresponse()
        assert r.status == 200
        r.read()
        c2 = http.client.HTTPConnection("127.0.0.1", port, timeout=30)                            # a second connection once the first closes
        c.close()
        c2.request("GET", "/query", headers={"Connection": "close"})
        r = c2.getresponse()
        assert r.status == 200 and r.getheader("Connection") == "close"
        r.read()
        c2.close()
        out, _ = p.communicate(timeout=30)
        assert f"served {n}" in out
    finally:
        if p.poll() is None:
            p.kill()


def test_http_listener_is_not_blocked_by_idle_or_stalled_clients():
    """one half-open socket must not deny service.  While an idle keep-alive connection and a connection that has sent
    half a request head AND one that has sent half a body stay open, a third client is served at once; the stalled ones then
    complete their requests on the same sockets and are served too (one poll set, no blocking recv)."""
    import http.client
    import socket
    import time
    n = 5
    p = subprocess.Popen([BIN, "http", str(n)], stdout=subprocess.PIPE, text=True)
    try:
        port = int(p.stdout.readline().split()[1])
        idle = http.client.HTTPConnection("127.0.0.1", port, timeout=30)
        idle.request("GET", "/query")                                   # request 1, then the connection idles (keep-alive)
        assert idle.getresponse().read()
        half_head = socket.create_connection(("127.0.0.1", port))
        half_head.sendall(b"POST /echo HTTP/1.1\r\nHost: x\r\nContent-Le")
        half_body = socket.create_connection(("127.0.0.1", port))
        half_body.sendall(b"POST /echo HTTP/1.1\r\nHost: x\r\nContent-Length: 10\r\n\r\n01234")
        time.sleep(0.3)
        t0 = time.time()
        c = http.client.HTTPConnection("127.0.0.1", port, timeout=10)
        c.request("POST", "/echo", body="hello")                         # request 2
        r = c.getresponse()
        assert r.status == 200 and r.read() == b"hello"
        assert time.time() - t0 < 2.0, "a stalled client kept the listener busy"
        half_head.sendall(b"ngth: 3\r\n\r\nabc")                         # request 3 completes on its own socket
        resp = b""
        while b"abc" not in resp:
            resp += half_head.recv(4096)
        assert resp.startswith(b"HTTP/1.1 200")
        half_body.sendall(b"56789")                                      # request 4
        resp = b""
        while b"0123456789" not in resp:
            resp += half_body.recv(4096)
        assert resp.startswith(b"HTTP/1.1 200")
        idle.request("GET", "/query")                                   # request 5 on the connection that idled meanwhile
        assert idle.getresponse().status == 200
        out, _ = p.communicate(timeout=30)
        assert f"served {n}" in out
        for s_ in (half_head, half_body):
            s_.close()
    finally:
        if p.poll() is None:
            p.kill()


def test_http_request_deadline_is_per_request_not_per_byte():
    """a client that trickles one byte at a time (never idle for long) must still get its 408 once the REQUEST is older than the
    timeout -- an inter-byte idle timer would let 256 such clients hold every connection slot.  Listener with a 1.5 s request timeout; the
    trickler sends a byte every 0.2 s.  Meanwhile a well-behaved client is served, and a keep-alive connection that waits LONGER than the
    request timeout between two complete requests is not punished for it (the deadline runs from a request's first byte)."""
    import http.client
    import socket
    import time
    p = subprocess.Popen([BIN, "http", "3", "1500"], stdout=subprocess.PIPE, text=True)
    try:
        port = int(p.stdout.readline().split()[1])
        keep = http.client.HTTPConnection("127.0.0.1", port, timeout=30)
        keep.request("POST", "/echo", body="one")                          # request 1
        assert keep.getresponse().read() == b"one"
        s_ = socket.create_connection(("127.0.0.1", port))
        head = b"POST /echo HTTP/1.1\r\nHost: x\r\nContent-Length: 400\r\n\r\n"
        s_.sendall(head)
        s_.settimeout(0.01)
        t0, resp = time.time(), b""
        while time.time() - t0 < 6.0 and b"\r\n\r\n" not in resp:
            try:
                s_.sendall(b"x")
            except OSError:
                break
            try:
                got = s_.recv(4096)
                if not got:
                    break
                resp += got
            except socket.timeout:
                pass
            time.sleep(0.2)
        waited = time.time() - t0
        assert resp.startswith(b"HTTP/1.1 408"), resp[:60]
        assert 1.2 < waited < 3.5, waited
        c = http.client.HTTPConnection("127.0.0.1", port, timeout=10)
        c.request("POST", "/echo", body="hello")                           # request 2
        assert c.getresponse().read() == b"hello"
        time.sleep(0.5)                                                    # (the keep-alive connection has now idled > 1.5 s since request 1)
        keep.request("POST", "/echo", body="two")                          # request 3
        r = keep.getresponse()
        assert r.status == 200 and r.read() == b"two"
        out, _ = p.communicate(timeout=30)
        assert "served 3" in out
        s_.close()
    finally:
        if p.poll() is None:
            p.kill()
