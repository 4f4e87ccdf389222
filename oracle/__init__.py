"""CPU oracle for the ANN hard-negative refresh path -- TEST INFRASTRUCTURE ONLY.

Everything under ``oracle/`` is a CPU restatement of the reference algorithm
(microsoft/ANCE ``drivers/run_ann_data_gen.py`` and what it calls).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  The product package ``ance_amd`` never does: it fails loudly when the
HIP library is missing instead of falling back to anything in here.

Pinning status (see DESIGN.md "Oracle"):

* encoder arithmetic  -- pinned against the reference's own classes
  (``model/models.py``) imported in the build container; the generating script
  ``tests/golden/make_golden.py`` and its vectors are committed.
* post-search logic   -- pinned against the reference's own functions
  (``GenerateNegativePassaageID``, ``EvalDevQuery``, ``generate_new_ann``) run
  through ``oracle/ref_harness.py`` in the build container; vectors committed.
* search (FAISS)      -- the arithmetic lives in ``faiss-cpu`` (unpinned in the
  reference's ``setup.py:22``, absent here).  The reference holds no golden
  vectors for it: **parity unpinned** at the FAISS boundary.  The oracle states
  the published algorithm (exact inner product, k largest, sorted descending)
  under a documented canonical total order (score desc, row-id asc) and an
  fp32 ``fmaf`` chain in ascending k so that ids are bit-reproducible.
"""
