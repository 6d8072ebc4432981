"""CPU oracle for the audio-cut separate+detect hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker.  The product package
(``audio_cut_amd``) never imports this package and fails loudly when its HIP
library is missing.

What it is: a plain numpy/scipy restatement of the reference's CPU algorithm
for SURVEY.md §8 rows a1-a18, each function citing the reference file:line it
follows (paths relative to /root/reference).

Pinning status
--------------
* ``refine`` (a17), ``chunk_schedule`` (a1), ``derive`` (a15), SileroChunkVAD
  merge/clip logic (a13) and the detector control logic (a14, a16) are pinned
  against the reference's own Python, imported in the build container by
  ``tests/golden/make_golden.py`` (fixtures committed under ``tests/golden``).
* The librosa / onnxruntime / silero_vad / MVSEP-MDX23 arithmetic is NOT in
  /root/reference and not installed: ``librosa_ops`` restates
  the published librosa 0.10 algorithms and is checked by closed-form
  known-answer tests only -> **parity unpinned** for those float series.
"""
