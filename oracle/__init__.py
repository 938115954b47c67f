"""TEST INFRASTRUCTURE ONLY: CPU restatement of the reference decode path (see po_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product package (poreover_amd) never does.
"""
