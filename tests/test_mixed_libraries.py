"""One process, the three libraries at once: what a JVM running GATK's HaplotypeCaller holds -- Smith-Waterman, PairHMM and
(in its partially-determined mode) PDHMM natives loaded side by side, each with its own device context, streams and pinned
buffers on the same GPU.  Threads call into all three concurrently for a few seconds (GKL_MIXED_SECONDS, default 4; a soak
sets it to minutes) while contexts of each library are opened and closed next to them; every answer is compared bit for bit
with the oracle's, computed once up front."""
import os
import threading
import time

import numpy as np
import pytest

from gkl_amd.synth import make_batch
from tests.test_pdhmm import cross_product, random_pd_batch
from tests.test_sw import PARAM_SETS, random_pairs


@pytest.mark.gpu
def test_three_libraries_side_by_side_in_one_process():
    from gkl_amd import native
    from oracle.oracle import Oracle
    from oracle.pdhmm import PdhmmOracle
    from oracle.sw import SOFTCLIP, SwOracle

    seconds = float(os.environ.get("GKL_MIXED_SECONDS", "4"))
    o, po, so = Oracle(), PdhmmOracle(), SwOracle()
    # PairHMM: GATK-sized regions (the combined small-call path), one mid-size and one two-pass call
    ph = [make_batch("hc", r, h, seed=40 + i) for i, (r, h) in enumerate([(100, 10), (37, 5), (260, 12), (900, 40), (2500, 64)])]
    ph_exp = [o.batch(b, n_threads=8) for b in ph]
    # PDHMM: paired batches (table route and not) and a cross product
    rng = np.random.RandomState(5)
    pd = [random_pd_batch(rng, 96), random_pd_batch(rng, 700, read_len=(40, 160), hap_len=(60, 260), with_n=False),
          cross_product(rng, 60, 5, (60, 151), (120, 260))]
    pd_exp = [po.compute(b, semantics=2)[1] for b in pd]
    # Smith-Waterman: a batch and single pairs
    pairs = random_pairs(np.random.RandomState(9), 200)
    stride = 2 * max(max(len(r), len(a)) for r, a in pairs)
    sw_exp = [so.align(r, a, PARAM_SETS[0], SOFTCLIP, cigar_len=stride) for r, a in pairs]
    sw_one = [so.align(r, a, PARAM_SETS[0], SOFTCLIP) for r, a in pairs[:24]]

    stop = time.time() + seconds
    errors, counts = [], {"pairhmm": 0, "pdhmm": 0, "sw": 0, "churn": 0}
    lock = threading.Lock()

    def guarded(name, body):
        def run():
            n = 0
            try:
                n = body()
            except BaseException as e:  # noqa: BLE001 -- reported by the main thread
                errors.append((name, repr(e)))
            with lock:
                counts[name.split("-")[0]] += n
        return threading.Thread(target=run, name=name)

    def pairhmm_caller(k):
        def body():
            n = 0
            with native.PairHmmContext(max_threads=2) as c:
                while time.time() < stop and not errors:
                    i = (n + k) % len(ph)
                    assert c.compute(ph[i]).tobytes() == ph_exp[i].tobytes(), ("pairhmm", k, i)
                    n += 1
            return n
        return body

    def pdhmm_caller(k):
        def body():
            n = 0
            with native.PdhmmContext(fma_mode=1, reference_tail=False) as c:
                while time.time() < stop and not errors:
                    i = (n + k) % len(pd)
                    assert c.compute(pd[i]).tobytes() == pd_exp[i].tobytes(), ("pdhmm", k, i)
                    n += 1
            return n
        return body

    def sw_caller(k):
        def body():
            n = 0
            with native.SwContext() as c:
                while time.time() < stop and not errors:
                    if (n + k) % 2 == 0:
                        cig, cnt, off = c.align_batch([r for r, _ in pairs], [a for _, a in pairs], PARAM_SETS[0], SOFTCLIP)
                        for j, (st, ecig, ecnt, eoff) in enumerate(sw_exp):
                            assert st == 0 and (cig[j], int(cnt[j]), int(off[j])) == (ecig, ecnt, eoff), ("sw batch", k, j)
                    else:
                        for j, (st, ecig, ecnt, eoff) in enumerate(sw_one):
                            assert c.align(pairs[j][0], pairs[j][1], PARAM_SETS[0], SOFTCLIP) == (ecig, ecnt, eoff), ("sw", k, j)
                    n += 1
            return n
        return body

    def churn():
        # contexts of every library come and go while the others are busy (initNative / doneNative of a second tool instance)
        n = 0
        while time.time() < stop and not errors:
            with native.PairHmmContext() as c:
                assert c.compute(ph[1]).tobytes() == ph_exp[1].tobytes()
            with native.PdhmmContext(fma_mode=1, reference_tail=False) as c:
                assert c.compute(pd[0]).tobytes() == pd_exp[0].tobytes()
            with native.SwContext() as c:
                st, ecig, ecnt, eoff = sw_one[0]
                assert c.align(pairs[0][0], pairs[0][1], PARAM_SETS[0], SOFTCLIP) == (ecig, ecnt, eoff)
            n += 1
        return n

    threads = ([guarded(f"pairhmm-{k}", pairhmm_caller(k)) for k in range(4)] + [guarded(f"pdhmm-{k}", pdhmm_caller(k)) for k in range(2)]
               + [guarded(f"sw-{k}", sw_caller(k)) for k in range(2)] + [guarded("churn-0", churn)])
    for t in threads:
        t.start()
    for t in threads:
        t.join(seconds + 120)
    assert not any(t.is_alive() for t in threads), "a caller never came back"
    assert not errors, errors
    assert all(v > 0 for v in counts.values()), counts
    print("mixed libraries:", counts, "calls bit-identical in", seconds, "s")
