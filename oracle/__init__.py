"""Oracle package: CPU restatement of the reference's GP predict path.

TEST INFRASTRUCTURE ONLY.  See gp_oracle.py for the rules on who may import it.
"""
