"""CPU oracle of the reference hot path - TEST INFRASTRUCTURE (see oracle/ngw_oracle.c header)."""
