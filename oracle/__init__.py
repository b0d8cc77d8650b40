"""CPU oracle for the vectordb search hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``vectordb_amd/`` may import this
package: it exists so that ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` can check (and time) the HIP path against
a NumPy restatement of the reference's arithmetic.
"""
