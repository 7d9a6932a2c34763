#!/usr/bin/env python3
"""Decoder-pool scaling on this host: JPEG files -> DecodePool only (no GPU), workers 1 / 2 / 4 / 8 / 16 / 32, pinned to
distinct physical cores first (decode_pool.cpu_order) and unpinned, warm.  Prints the host's topology next to the rates.
    python tools/decode_sweep.py [files]"""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from PIL import Image

from hse_facerec_tf_amd.decode_pool import DecodePool, cpu_order

B = 256


def rates(nw, files, pin, task_files=8):
    pool = DecodePool(nw, slot_bytes=max(8 << 20, B * (256 << 10)), slots=3, task_files=task_files, pin=pin)
    try:
        chunks = [files[i:i + B] for i in range(0, len(files), B)]

        def one():
            t0 = time.perf_counter()
            for ci in range(min(2, len(chunks))):
                pool.submit(ci, chunks[ci], ci % 3)
            for ci in range(len(chunks)):
                pool.collect(ci)
                if ci + 2 < len(chunks):
                    pool.submit(ci + 2, chunks[ci + 2], (ci + 2) % 3)
            return len(files) / (time.perf_counter() - t0)
        one()
        return max(one() for _ in range(2))
    finally:
        pool.close()


def main():
    nfiles = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    order = cpu_order()
    sib = set()
    for c in order:
        try:
            sib.add(open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c).read().strip())
        except OSError:
            sib.add(str(c))
    print("cpus allowed %d, physical cores %d, order %s" % (len(order), len(sib), order))
    try:
        print("loadavg", open("/proc/loadavg").read().strip())
    except OSError:
        pass
    rs = np.random.RandomState(1)
    d = tempfile.mkdtemp(prefix="hsefr_sweep_")
    try:
        for i in range(256):
            Image.fromarray(rs.randint(0, 256, (250, 250, 3), dtype=np.uint8)).save(os.path.join(d, "%04d.jpg" % i), quality=90)
        paths = [os.path.join(d, "%04d.jpg" % (i % 256)) for i in range(nfiles)]
        for nw in (1, 2, 4, 8, 16, 32):
            if nw > len(order):
                break
            n = min(nfiles, max(512, 256 * nw))
            rp = rates(nw, paths[:n], True)
            ru = rates(nw, paths[:n], False)
            print("%2d workers: pinned %7.0f img/s (%5.0f per worker) | unpinned %7.0f (%5.0f per worker)" % (nw, rp, rp / nw, ru, ru / nw), flush=True)
        # where does it stop scaling?  other runs of CPUs (the first ones of a shared host are everybody's favourite), more workers
        for off in (32, 64, 96):
            if off + 32 <= len(order):
                os.environ["HSEFR_DECODE_CPU_OFFSET"] = str(off)
                r = rates(32, paths[:min(nfiles, 8192)], True)
                print("32 workers on cpus %d..%d: %7.0f img/s (%5.0f per worker)" % (off, off + 31, r, r / 32), flush=True)
        os.environ["HSEFR_DECODE_CPU_OFFSET"] = "0"
        for nw in (48, 64, 96):
            if nw <= len(order):
                r = rates(nw, paths[:min(nfiles, 8192)], True)
                print("%2d workers: pinned %7.0f img/s (%5.0f per worker)" % (nw, r, r / nw), flush=True)
        nw = min(32, len(order))
        for tf in (4, 16):
            r = rates(nw, paths[:min(nfiles, 256 * nw)], True, task_files=tf)
            print("%2d workers, %2d files per task: %7.0f img/s" % (nw, tf, r), flush=True)
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
