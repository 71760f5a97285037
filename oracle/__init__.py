"""CPU oracle (test infrastructure). See oracle/tr_oracle.h."""
