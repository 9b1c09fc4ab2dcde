SC_PHI_PROFILE=1 SC_LIB=spatialcore_amd/libvar_phiprofile.so timeout -k 10 120 python3 scripts/generator_probe.py 1000000 1000 2>&1 | grep "^rounds\|^fixed\|x 1000000"
