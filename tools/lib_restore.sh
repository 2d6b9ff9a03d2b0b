# Sourced by the scripts that rebuild csrc/ IN-TREE with non-product flags (STAMPS=1, DEBUG_SWITCHES=1, EXPERIMENTS=1, -D knobs):
# saves the product libsed_hip.so and installs an EXIT trap that rebuilds the product flags and compares the result with the saved
# file, so that a failure or an interrupt in the middle cannot leave a stamped / ablated library in the tree.
#   source tools/lib_restore.sh      (from the repository root, before the first non-product make)
SED_ROOT=$(pwd)
SED_KEEP=$(mktemp /tmp/libsed_hip.so.keep.XXXXXX)
cp "$SED_ROOT/soundeventdetection-pytorch_amd/libsed_hip.so" "$SED_KEEP"
sed_restore_product_build() {
    local rc=$?
    # SED_NO_RESTORE=1: a throw-away snapshot (gpurun copies the tree to a fresh box and discards it): skip the ~75 s rebuild
    if [ -n "$SED_NO_RESTORE" ]; then rm -f "$SED_KEEP"; exit $rc; fi
    cd "$SED_ROOT/soundeventdetection-pytorch_amd/csrc" && rm -f *.o && make -j14 > /tmp/mk_restore.log 2>&1 || { echo "RESTORE BUILD FAILED (see /tmp/mk_restore.log)"; exit 1; }
    if cmp -s "$SED_ROOT/soundeventdetection-pytorch_amd/libsed_hip.so" "$SED_KEEP"; then echo "product build restored (byte-identical)"; else echo "product build restored, but it DIFFERS from the library found at start"; fi
    rm -f "$SED_KEEP"
    exit $rc
}
trap sed_restore_product_build EXIT
