"""CPU oracle for the dual-QP hot path: TEST INFRASTRUCTURE ONLY (see bcqp_oracle.py / svm_oracle.py)."""
