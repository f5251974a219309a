"""ORACLE -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package, and only as the checker / the timed CPU baseline.  The product package
(p_companion_amd) never imports it and has no CPU fallback.
"""
