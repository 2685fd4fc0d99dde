"""CPU oracle package -- test infrastructure only (see ppo_oracle.py header)."""
