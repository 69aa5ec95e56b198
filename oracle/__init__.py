"""CPU oracle for the MFCC + Gaussian-HMM hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package imports this
directory; it is used by ``tests/``, by ``__graft_entry__.smoke()`` and by the
``cpu_baseline`` leg of ``bench.py`` as the checker / CPU timing baseline.

Parity status: PINNED.  Every function here is checked against outputs of the
reference's own modules (``mfcc.py``, ``my_hmmlearn/hmm.py``,
``my_hmmlearn/utils.py``) that were produced by importing them in the build
container; the vectors live in ``tests/golden/*.npz`` and the script that made
them is ``tests/golden/make_golden.py``.
"""
