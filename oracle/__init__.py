"""CPU oracle for the RAN-slice hot path.  TEST INFRASTRUCTURE -- see rs_oracle.h."""
