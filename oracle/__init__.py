"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md). Never imported by bonito_amd/."""
