#!/bin/sh
# Adds the MI355X binding to a checkout of xelis-he by appending to its files and copying new ones (see README.md).
set -e
: "${XELIS_HE:?path of the xelis-he checkout}" "${XHE_REPO:?path of this repository}"
here="$XHE_REPO/bindings/rust"
cp "$here/src/gpu.rs" "$XELIS_HE/src/tx/gpu.rs"
cp "$here/build.rs" "$XELIS_HE/build.rs"
# the tests are in-crate modules: `tx`, VerificationError and the compressed point types are private to the crate
cp "$here/tests/ab.rs" "$XELIS_HE/src/tx/gpu_ab.rs"
cp "$here/tests/golden.rs" "$XELIS_HE/src/tx/gpu_golden.rs"
printf '\n#[cfg(feature = "mi355x")]\npub mod gpu;\n#[cfg(all(test, feature = "mi355x"))]\nmod gpu_ab;\n#[cfg(all(test, feature = "mi355x"))]\nmod gpu_golden;\n' >> "$XELIS_HE/src/tx/mod.rs"
cat "$here/patch/proofs_from_bytes.rs" >> "$XELIS_HE/src/proofs.rs"
cat "$here/patch/extra_data_from_bytes.rs" >> "$XELIS_HE/src/extra_data.rs"
# the manifest ends with [dev-dependencies]: the first line lands there, the feature table follows
printf 'serde_json = "1"\n\n[features]\nmi355x = []\n' >> "$XELIS_HE/Cargo.toml"
echo "applied; build with: XHE_LIB_DIR=$XHE_REPO/xelis-he_amd cargo test --features mi355x gpu_ -- --test-threads 1"
