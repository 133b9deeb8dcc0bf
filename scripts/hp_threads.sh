for T in 8 16 32 64; do echo "== ZN_HOST_THREADS=$T"; ZN_HOST_THREADS=$T python scripts/host_path_check.py 2>&1 | grep -E "one shot|automatic|4 slices" ; done
