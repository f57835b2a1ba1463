//! Placeholder root: `gpu.rs` is a module of the xelis-he crate (`src/tx/gpu.rs` after `apply.sh`), not of this package.
//! See README.md.
