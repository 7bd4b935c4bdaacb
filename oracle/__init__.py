"""oracle/ — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product
package under gym.net_amd/ never imports this.  PARITY UNPINNED for CartPole (see the header of
classic_control_ref.c); Philox4x32-10 is pinned by the Random123 known-answer vectors.
"""
