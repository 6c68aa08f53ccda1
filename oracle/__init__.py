"""CPU oracle package (test infrastructure only - see oracle/quatro_oracle.cpp header)."""
