"""GPU: the posterior / genotype-call kernels and the stutter EM at their size boundaries, randomised (tools/fuzz_post.py, tools/fuzz_em.py):
allele counts around a wavefront's 64 lanes and its multiples, diplotype counts around the 2048 a unit keeps in registers, samples
without reads, one sample ... hundreds, weights 0, haploid loci, iteration caps.  Contract: every output equals the oracle evaluated
with the same correctly rounded exp / log bit for bit (level 2 — that run is itself held to the host-libm reference by the
other suites).  A short run of each here; the long runs are profiles/r05_fuzz_boundaries.txt."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_posteriors_and_genotype_calls_at_size_boundaries(hmm, oracle):
    import fuzz_post
    bad, units = fuzz_post.run(24, 7, hmm, oracle)
    assert bad == 0 and units > 100000


def test_stutter_em_at_size_boundaries(hmm, oracle):
    import fuzz_em
    bad, loci = fuzz_em.run(5, 7, hmm, oracle)
    assert bad == 0 and loci > 0
