import subprocess, sys
src = open('../../rustpotter_amd/csrc/rp_kernels.hip').read()
exp = open('mfcc_exp.hip').read()
variants = {
 'base': [],
 'noload': [("const float cur = g < n_samples ? x[g] : 0.f;", "const float cur = (float)(i & 255) * 0.001f;"),
            ("const float prev = (i % kShift == 0 || g >= n_samples) ? 0.f : x[g - 1];", "const float prev = (float)((i-1) & 255) * 0.001f;")],
 'nostore': [("dst[c - 1] = 2.f * sum;", "if (sum == 1.2345f) dst[c - 1] = 2.f * sum;")],
 'nolog': [("lgb[i0 + ii] = logf(tot + FLT_MIN);", "lgb[i0 + ii] = tot + FLT_MIN;")],
 'nomel': [("acc[ii] = fmaf(P[k2], row[16 * k2], acc[ii]);", "acc[ii] += P[k2];")],
 'nofft': [("        fft16(v);\n", "")],
 'nodft15': [("            dft15(u, z);", "#pragma unroll\n for (int q_=0;q_<15;++q_) z[q_]=u[q_];")],
}
for name, reps in variants.items():
    s = src
    for a, b in reps:
        assert a in s, (name, a)
        s = s.replace(a, b)
    open('k_%s.hip' % name, 'w').write(s)
    e = exp.replace('#include "../../rustpotter_amd/csrc/rp_kernels.hip"', '#include "k_%s.hip"' % name)
    e = e.replace('#include "../../rustpotter_amd/csrc/rp_host.h"', '#include "../../rustpotter_amd/csrc/rp_host.h"')
    open('e_%s.hip' % name, 'w').write(e)
    subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off', '-I../../rustpotter_amd/csrc',
                           'e_%s.hip' % name, '../../rustpotter_amd/csrc/rp_tables.cpp', '-o', 'exp_%s' % name])
print('built', list(variants))
