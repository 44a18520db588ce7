"""CPU oracle for the RNN-T loss path (test infrastructure; never imported by rnnt_amd)."""
