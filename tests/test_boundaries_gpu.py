"""GPU: randomised sweeps at the kernels' size boundaries (tools/fuzz_post.py, fuzz_em.py, fuzz_misc.py, fuzz_mixed.py).  Posteriors, genotype calls, EM:
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


def _keep_generator_settings(monkeypatch):
    """The fuzzers set the generator's HIPSTR_SYNTH_* overrides in os.environ as they go: registered with monkeypatch first, the variables are
    back to what they were (normally: unset) when the test ends — later tests and the processes they start see the default generator."""
    for k in ("HIPSTR_SYNTH_IMPERFECT", "HIPSTR_SYNTH_INHERIT"):
        was_set = k in os.environ
        monkeypatch.setenv(k, os.environ.get(k, "0"))
        if not was_set:
            monkeypatch.delenv(k)


def test_needleman_wunsch_and_caller_chosen_seeds_at_size_boundaries(hmm, oracle, monkeypatch):
    """tools/fuzz_misc.py: Needleman-Wunsch on reads of 1 ... 1536 bases against windows of 1 ... 3000 around the kernel's tile sizes; the
    forward path and the traceback with seeds anywhere in the read (HapAligner.h:83, :93: the seed base is the caller's argument)."""
    import fuzz_misc
    _keep_generator_settings(monkeypatch)
    assert fuzz_misc.run(12, 7, hmm, oracle) == 0


def test_heterogeneous_batches_in_one_call_and_through_the_stream(hmm, oracle, monkeypatch):
    """tools/fuzz_mixed.py: loci that have nothing in common (read and flank lengths, allele counts, periods, interrupted or plain repeats,
    masks) concatenated into one batch — what the host pipeline's batches look like in production (bam_processor.cpp:550-617: one region
    after the other) — in one process_reads call and one locus per submission through the stream, against the oracle."""
    import fuzz_mixed
    _keep_generator_settings(monkeypatch)
    assert fuzz_mixed.run(6, 7, hmm, oracle, stream_every=2) == 0
