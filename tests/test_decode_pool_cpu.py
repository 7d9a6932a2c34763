"""CPU suite: the decoder-process pool behind extract_files (decode_pool.py) -- same pixels as the serial
``preprocess.imread_rgb`` (misc.imread(mode='RGB'), facerec_test.py:83), packing, spills, error propagation, clean shutdown."""
import os

import numpy as np
import pytest
from PIL import Image

from hse_facerec_tf_amd import preprocess
from hse_facerec_tf_amd.decode_pool import DecodePool


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    d = tmp_path_factory.mktemp("imgs")
    rs = np.random.RandomState(11)
    paths = []
    for i in range(37):
        hw = (90, 70) if i % 5 else (64, 128)
        mode_gray = (i == 9)                                     # a grayscale file: convert('RGB') must triple it
        a = rs.randint(0, 256, hw if mode_gray else hw + (3,), dtype=np.uint8)
        p = str(d / ("%02d.%s" % (i, "png" if i % 2 else "jpg")))
        Image.fromarray(a).save(p)
        paths.append(p)
    return paths


def _pixels(pool, m, sp):
    if m is None:
        return sp
    buf = np.frombuffer(pool._shm.buf, dtype=np.uint8)
    return buf[m[0]:m[0] + m[1] * m[2] * 3].reshape(m[1], m[2], 3).copy()


def test_pool_decodes_what_imread_rgb_decodes_and_packs_back_to_back(files):
    pool = DecodePool(2, slot_bytes=4 << 20, slots=2, task_files=8)
    try:
        pool.submit("a", files, 0)
        pool.submit("b", files[::-1], 1)                         # two chunks in flight, collected out of order
        rb = pool.collect("b")
        ra = pool.collect("a")
        for res, paths, slot in ((ra, files, 0), (rb, files[::-1], 1)):
            assert [pos for pos, _, _ in res] == list(range(len(paths)))
            for pos, m, sp in res:
                assert m is not None and slot * (4 << 20) <= m[0] < (slot + 1) * (4 << 20)
                assert np.array_equal(_pixels(pool, m, sp), preprocess.imread_rgb(paths[pos])), paths[pos]
            # inside a task (8 files) the images are contiguous: a same-size run is ONE upload
            for t in range(0, len(paths), 8):
                offs = [res[k][1] for k in range(t, min(t + 8, len(paths)))]
                for (o0, h0, w0), (o1, _, _) in zip(offs, offs[1:]):
                    assert o1 == o0 + h0 * w0 * 3
    finally:
        pool.close()
    assert pool._shm is None


def test_images_that_do_not_fit_come_back_through_the_queue_and_errors_propagate(files, tmp_path):
    pool = DecodePool(2, slot_bytes=64 << 10, slots=1, task_files=8)       # 64 KiB per slot: most images spill
    try:
        pool.submit(0, files[:16], 0)
        res = pool.collect(0)
        assert sum(m is None for _, m, _ in res) > 0 and sum(m is not None for _, m, _ in res) > 0
        for pos, m, sp in res:
            assert np.array_equal(_pixels(pool, m, sp), preprocess.imread_rgb(files[pos]))
        pool.submit(1, files[:3] + [str(tmp_path / "nope.jpg")], 0)
        with pytest.raises(FileNotFoundError):
            pool.collect(1)
        bad = tmp_path / "broken.jpg"
        bad.write_bytes(b"not an image")
        pool.submit(2, [str(bad)], 0)
        with pytest.raises(Exception) as ei:
            pool.collect(2)
        with pytest.raises(Exception) as serial:
            preprocess.imread_rgb(str(bad))
        # the SAME exception type as the serial path (PIL.UnidentifiedImageError: not a builtin, an OSError callers may catch)
        assert type(ei.value) is type(serial.value) and isinstance(ei.value, OSError)
        pool.submit(3, files[:5], 0)                                        # still alive after errors
        assert len(pool.collect(3)) == 5
    finally:
        pool.close()
    pool.close()                                                            # idempotent


def test_worker_errors_rebuild_in_the_parent_and_workers_share_the_node(monkeypatch):
    """non-builtin exception types and builtins whose constructors want several arguments must not turn into a
    TypeError in collect(); VERDICT r3 #6: the default decoder count is divided between the ranks of a node."""
    import pickle
    from hse_facerec_tf_amd import decode_pool
    from PIL import UnidentifiedImageError
    e = decode_pool._rebuild_error("UnidentifiedImageError", "cannot identify image file 'x'", "x",
                                   pickle.dumps(UnidentifiedImageError("cannot identify image file 'x'")))
    assert type(e) is UnidentifiedImageError
    e = decode_pool._rebuild_error("UnicodeDecodeError", "codec can't decode", "f.jpg", None)      # 5-argument constructor
    assert type(e) is RuntimeError and "UnicodeDecodeError" in str(e) and "f.jpg" in str(e)
    e = decode_pool._rebuild_error("FileNotFoundError", "gone", "/x/y.jpg", None)
    assert type(e) is FileNotFoundError and e.filename == "/x/y.jpg"
    assert type(decode_pool._rebuild_error("ValueError", "bad", "p", b"not a pickle")) is ValueError
    monkeypatch.setattr(decode_pool.os, "sched_getaffinity", lambda _pid: set(range(64)), raising=False)
    monkeypatch.setattr(decode_pool, "cpu_quota", lambda: None)
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    assert decode_pool.default_workers() == 32
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert decode_pool.default_workers() == 8                               # 64 cores / 8 ranks: 64 decoders on the host, not 256
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "128")
    assert decode_pool.default_workers() == 1
    # round 5: the container's CPU quota caps the pool (the GPU boxes list 256 CPUs under a 16-CPU cgroup quota: 32 decoders decoded
    # what 16 do), and the workers pin themselves to distinct physical cores first
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    monkeypatch.setattr(decode_pool, "cpu_quota", lambda: 16.0)
    assert decode_pool.default_workers() == 16
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    assert decode_pool.default_workers() == 2
    q = decode_pool.cpu_quota.__wrapped__() if hasattr(decode_pool.cpu_quota, "__wrapped__") else None
    assert q is None or q > 0
    order = decode_pool.cpu_order()
    assert sorted(order) == sorted(set(order)) and len(order) >= 1


def test_pinning_is_opt_in_unless_a_launcher_says_where_the_process_is(monkeypatch):
    from hse_facerec_tf_amd import decode_pool
    """ADVICE r5: processes started by hand have no LOCAL_RANK -- pinning them all to cpu_order()[0:workers] would stack every
    job's decoders on the same cores.  Default: pin only when LOCAL_RANK / SLURM_LOCALID / HSEFR_DECODE_CPU_OFFSET is set; pools
    alive in one process take consecutive runs of the list; a closed pool gives its run back."""
    for k in ("LOCAL_RANK", "SLURM_LOCALID", "HSEFR_DECODE_CPU_OFFSET"):
        monkeypatch.delenv(k, raising=False)
    assert decode_pool.pin_offset(4) is None
    monkeypatch.setenv("SLURM_LOCALID", "3")
    assert decode_pool.pin_offset(4) == 12
    monkeypatch.setenv("LOCAL_RANK", "2")                      # torchrun wins over srun's variable when both are set
    assert decode_pool.pin_offset(4) == 8
    monkeypatch.setenv("HSEFR_DECODE_CPU_OFFSET", "5")
    assert decode_pool.pin_offset(4) == 13
    for k in ("LOCAL_RANK", "SLURM_LOCALID", "HSEFR_DECODE_CPU_OFFSET"):
        monkeypatch.delenv(k, raising=False)
    order = decode_pool.cpu_order()
    a = DecodePool(1, slot_bytes=1 << 20, slots=1)
    try:
        assert a.cpus == [None]                                  # nothing known about the node: the scheduler places the worker
    finally:
        a.close()
    monkeypatch.setenv("LOCAL_RANK", "1")
    a = DecodePool(1, slot_bytes=1 << 20, slots=1)
    b = DecodePool(1, slot_bytes=1 << 20, slots=1)
    try:
        assert a.cpus == [order[1 % len(order)]] and b.cpus == [order[2 % len(order)]]
        a.close()
        c = DecodePool(1, slot_bytes=1 << 20, slots=1, pin=True)
        try:
            assert c.cpus == a.cpus                              # the released run is taken again
        finally:
            c.close()
    finally:
        a.close()
        b.close()
    monkeypatch.delenv("LOCAL_RANK")
    d = DecodePool(1, slot_bytes=1 << 20, slots=1, pin=True)   # forced: offset 0
    try:
        assert d.cpus == [order[0]]
    finally:
        d.close()
