"""CPU oracle (test infrastructure only — see oracle/mld_oracle.cpp)."""
