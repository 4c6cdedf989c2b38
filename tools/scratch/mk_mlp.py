import subprocess
src = open('../../rustpotter_amd/csrc/rp_kernels.hip').read()
exp = open('mlp_exp.hip').read()
variants = {
 'base': [],
 'noepi': [("    if (l < kMlpRowsPerWave && row0 + l < B) {\n        const float *hin = h1", "    if (l < kMlpRowsPerWave && row0 + l < B) { out[(row0 + l) * 2] = h1[l * (N1P + 1)]; }\n    if (false) {\n        const float *hin = h1")],
 'nob': [("for (int n = 0; n < NT; ++n) b[u][n] = *reinterpret_cast<const float4 *>(w1f + (size_t)(16 * n + li) * kpad + k0);", "for (int n = 0; n < NT; ++n) b[u][n] = make_float4(1.f, 2.f, 3.f, (float)k0);"),
         ("for (int n = 0; n < NT; ++n) b[u][n] = *reinterpret_cast<const bf16x8 *>(w1h + (size_t)(16 * n + li) * kpad + k0);", "for (int n = 0; n < NT; ++n) { bf16x8 q_; for (int e_ = 0; e_ < 8; ++e_) q_[e_] = (__bf16)(float)(k0 + e_); b[u][n] = q_; }")],
}
for name, reps in variants.items():
    s = src
    for a, b in reps:
        assert a in s, (name, a[:50])
        s = s.replace(a, b)
    open('km_%s.hip' % name, 'w').write(s)
    open('em_%s.hip' % name, 'w').write(exp.replace('KERNELS', 'km_%s.hip' % name))
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-I../../rustpotter_amd/csrc', 'em_%s.hip' % name,
                           '../../rustpotter_amd/csrc/rp_ctx.cpp', '../../rustpotter_amd/csrc/rp_tables.cpp', '-o', 'mexp_%s' % name], stderr=subprocess.DEVNULL)
print("built")
