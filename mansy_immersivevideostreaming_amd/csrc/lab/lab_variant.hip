// LAB BUILD ONLY (build_ext.build(lab=True) -> libmansy_hip_lab.so; never part of libmansy_hip.so): a settable default for the kernel-selection
// variant of calls that pass 0 (mansy_kernels.h: mansy_variant_of), so that whole engine steps can be A/B-timed on two loops (tools/lab_knobs.py).
// The release library has no process-wide mutable state and does not compile this file.
int g_mansy_lab_variant = 0;
extern "C" int mansy_lab_set_variant(int v) { const int old = g_mansy_lab_variant; if (v >= 0) g_mansy_lab_variant = v; return old; }
