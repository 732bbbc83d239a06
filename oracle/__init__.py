"""TEST INFRASTRUCTURE ONLY.

CPU oracle for the OpenDrift particle-advection hot path.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package; the product path (opendrift_amd) never does.
"""
