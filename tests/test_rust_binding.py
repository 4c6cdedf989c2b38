"""The Rust side of the boundary cannot be compiled in this image (no rustc), so this test stands in for the compiler's
FFI check: it parses include/rustpotter_hip.h and bindings/rustpotter_hip.rs and compares every function (name, argument
count, argument and return types), every #[repr(C)] struct (field order and types) and every enum constant.  It also checks
that the wrapper keeps the reference's public signatures (src/detector.rs:95-302)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_SCALARS = {
    "int": "c_int", "size_t": "usize", "float": "f32", "double": "f64", "bool": "bool", "char": "c_char", "void": "c_void",
    "uint8_t": "u8", "uint16_t": "u16", "uint64_t": "u64", "int8_t": "i8", "int16_t": "i16", "int32_t": "i32", "int64_t": "i64",
    "long long": "i64",
}


def _strip_c(src):
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    src = re.sub(r"^\s*#[^\n]*", "", src, flags=re.M)   # preprocessor lines
    src = src.replace('extern "C" {', "")
    return src


def _strip_rs(src):
    return re.sub(r"//[^\n]*", "", src)


def parse_header():
    src = _strip_c(open(os.path.join(ROOT, "include", "rustpotter_hip.h")).read())
    enums, consts = set(), {}
    for body, name in re.findall(r"typedef\s+enum\s*\{([^}]*)\}\s*(\w+)\s*;", src):
        enums.add(name)
        _enum_values(body, consts)
    for body in re.findall(r"(?<!typedef )\benum\s*\{([^}]*)\}\s*;", src):
        _enum_values(body, consts)
    opaque = set(re.findall(r"typedef\s+struct\s+(\w+)\s+\1\s*;", src))
    structs = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{([^}]*)\}\s*(\w+)\s*;", src):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            # "size_t epochs, test_epochs" declares two fields of one type
            base, names = _split_decl(decl)
            for n in names:
                fields.append((n, base))
        structs[name] = fields
    no_typedefs = re.sub(r"typedef\s+(struct|enum)\s*\{[^}]*\}\s*\w+\s*;", "", src)
    funcs = {}
    for ret, name, args in re.findall(r"([\w][\w\s\*]*?)\b(rp_\w+)\s*\(([^()]*)\)\s*;", no_typedefs):
        ret = " ".join(ret.split())
        params = []
        if args.strip() and args.strip() != "void":
            for a in args.split(","):
                base, names = _split_decl(" ".join(a.split()))
                assert len(names) == 1, (name, a)
                params.append((names[0], base))
        funcs[name] = (ret, params)
    return enums, consts, opaque, structs, funcs


def _enum_values(body, consts):
    nxt = 0
    for item in body.split(","):
        item = item.strip()
        if not item:
            continue
        if "=" in item:
            k, v = [x.strip() for x in item.split("=")]
            nxt = int(v, 0)
        else:
            k = item
        consts[k] = nxt
        nxt += 1


def _split_decl(decl):
    """'const float *const *weights' -> ('const float *const *', ['weights']);  'size_t epochs, test_epochs' -> two names"""
    first, *rest = [d.strip() for d in decl.split(",")]
    m = re.match(r"^(.*?)(\w+)$", first)
    base, name = m.group(1).strip(), m.group(2)
    return base, [name] + rest


def c_to_rust(ctype, enums, known):
    """canonical Rust spelling of a C type as the header writes it"""
    t = ctype.replace("*", " * ")
    toks = t.split()
    # pointer levels, right to left: each '*' may be followed by 'const' (qualifying the pointer itself, irrelevant to FFI
    # except for the pointee of the NEXT level out), preceded by the pointee's own const
    # parse: [const] base [const] (* [const])*
    i = 0
    const_base = False
    if toks[i] == "const":
        const_base = True
        i += 1
    if toks[i] == "long" and i + 1 < len(toks) and toks[i + 1] == "long":
        base = "long long"
        i += 2
    else:
        base = toks[i]
        i += 1
    if i < len(toks) and toks[i] == "const":
        const_base = True
        i += 1
    if base in enums:
        cur = "c_int"
    elif base in C_SCALARS:
        cur = C_SCALARS[base]
    else:
        assert base in known, "unknown C type %r in %r" % (base, ctype)
        cur = base
    pointee_const = const_base
    while i < len(toks):
        assert toks[i] == "*", ctype
        i += 1
        cur = ("*const " if pointee_const else "*mut ") + cur
        pointee_const = False
        if i < len(toks) and toks[i] == "const":
            pointee_const = True
            i += 1
    return cur


def norm_rust(t):
    t = " ".join(t.split())
    t = t.replace("std::ffi::c_void", "c_void").replace("std::os::raw::", "").replace("c_longlong", "i64")
    return t


def parse_rust():
    src = _strip_rs(open(os.path.join(ROOT, "bindings", "rustpotter_hip.rs")).read())
    consts = {k: int(v, 0) for k, v in re.findall(r"pub const (RP_\w+): c_int = (-?\w+);", src)}
    structs = {}
    for name, body in re.findall(r"#\[repr\(C\)\][^\n]*?pub struct (\w+) \{([^}]*)\}", src):
        structs[name] = [(n, norm_rust(t)) for n, t in re.findall(r"pub (\w+): ([^,}]+?)\s*(?:,|$)", body.strip())]
    opaque = set(re.findall(r"pub enum (\w+) \{\}", src))
    funcs = {}
    for block in re.findall(r'extern "C" \{(.*?)\n\}', src, flags=re.S):
        for name, args, ret in re.findall(r"pub fn (\w+)\(([^)]*)\)\s*(?:->\s*([^;]+))?;", block):
            params = []
            for a in args.split(","):
                a = a.strip()
                if a:
                    n, t = a.split(":", 1)
                    params.append((n.strip(), norm_rust(t)))
            funcs[name] = (norm_rust(ret) if ret else "", params)
    return consts, opaque, structs, funcs, src


def test_every_header_function_is_declared_with_the_same_signature():
    enums, _, opaque, structs, cfuncs = parse_header()
    _, ropaque, rstructs, rfuncs, _ = parse_rust()
    known = opaque | set(structs)
    decl = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "rustpotter_hip.h")).read(), flags=re.S)
    assert set(cfuncs) == set(re.findall(r"\b(rp_[a-z0-9_]+)\s*\(", decl)), "the prototype parser missed a declaration"
    missing = sorted(set(cfuncs) - set(rfuncs))
    extra = sorted(set(rfuncs) - set(cfuncs))
    assert not missing, "header functions without a Rust declaration: %s" % missing
    assert not extra, "Rust declarations without a header function: %s" % extra
    assert opaque == ropaque, (opaque, ropaque)
    bad = []
    for name, (cret, cparams) in sorted(cfuncs.items()):
        rret, rparams = rfuncs[name]
        want_ret = "" if cret == "void" else c_to_rust(cret, enums, known)
        if want_ret != rret:
            bad.append("%s: returns %r in C = %r, Rust says %r" % (name, cret, want_ret, rret))
        if len(cparams) != len(rparams):
            bad.append("%s: %d parameters in C, %d in Rust" % (name, len(cparams), len(rparams)))
            continue
        for i, ((cn, ct), (rn, rt)) in enumerate(zip(cparams, rparams)):
            want = c_to_rust(ct, enums, known)
            if want != rt:
                bad.append("%s arg %d (%s): C %r = %r, Rust says %r" % (name, i, cn, ct, want, rt))
    assert not bad, "\n".join(bad)


def test_repr_c_structs_have_the_headers_field_order_and_types():
    enums, _, opaque, structs, _ = parse_header()
    _, _, rstructs, _, _ = parse_rust()
    known = opaque | set(structs)
    assert set(structs) == set(rstructs), (sorted(structs), sorted(rstructs))
    bad = []
    for name, fields in structs.items():
        want = [(n, c_to_rust(t, enums, known)) for n, t in fields]
        if want != rstructs[name]:
            bad.append("%s:\n  header %s\n  rust   %s" % (name, want, rstructs[name]))
    assert not bad, "\n".join(bad)


def test_enum_constants_agree():
    _, consts, _, _, _ = parse_header()
    rconsts, _, _, _, _ = parse_rust()
    assert consts == rconsts, sorted(set(consts.items()) ^ set(rconsts.items()))


def test_wrapper_keeps_the_reference_signatures():
    """src/detector.rs:95-302: the methods a caller of `rustpotter::Rustpotter` uses, spelled as the reference spells them."""
    src = " ".join(parse_rust()[4].split())
    for sig in (
        "pub fn new(config: &RustpotterConfig) -> Result<Rustpotter, String>",
        "pub fn add_wakeword_ref(&mut self, key: &str, wakeword: WakewordRef) -> Result<(), String>",
        "pub fn add_wakeword_model(&mut self, key: &str, wakeword: WakewordModel) -> Result<(), String>",
        "pub fn add_wakeword_from_buffer(&mut self, key: &str, buffer: &[u8]) -> Result<(), String>",
        "pub fn add_wakeword_from_file(&mut self, key: &str, path: &str) -> Result<(), String>",
        "pub fn remove_wakeword(&mut self, key: &str) -> bool",
        "pub fn remove_wakewords(&mut self) -> bool",
        "pub fn get_samples_per_frame(&self) -> usize",
        "pub fn get_bytes_per_frame(&self) -> usize",
        "pub fn get_partial_detection(&self) -> Option<&RustpotterDetection>",
        "pub fn get_rms_level(&self) -> f32",
        "pub fn get_gain(&self) -> f32",
        "pub fn get_rms_level_ref(&self) -> f32",
        "pub fn process_bytes(&mut self, audio_bytes: &[u8]) -> Option<RustpotterDetection>",
        "pub fn process_samples<T: Sample>(&mut self, audio_samples: Vec<T>) -> Option<RustpotterDetection>",
        "pub fn update_config(&mut self, config: &RustpotterConfig)",
        "pub fn update_detector_config(&mut self, config: &DetectorConfig)",
        "pub fn update_filters_config(&mut self, config: &FiltersConfig)",
        "pub fn reset(&mut self)",
        "impl From<&RustpotterConfig> for rp_config",
        "impl From<&DetectorConfig> for rp_detector_config",
        "impl From<&FiltersConfig> for rp_filters_config",
        # the operator seam
        "pub fn mfcc_batch(&self, pcm: &[f32], n_streams: usize, n_samples: usize, mfcc_size: u16) -> Result<Vec<f32>, String>",
        "pub fn dtw_score_batch(&self, mfcc: &[f32], n_streams: usize, n_frames: usize, t: &Templates, score_ref: f32, band_size: u16, score_mode: ScoreMode, with_avg: bool) -> Result<WindowScores, String>",
        "pub fn mlp_forward_batch(&self, m: &Model, x: &[f32], rows: usize, bf16: bool) -> Result<Vec<f32>, String>",
    ):
        assert sig in src, sig


def test_every_extern_function_is_used_by_a_safe_wrapper():
    """a declaration nobody calls is dead weight a maintainer cannot trust: each rp_* function appears in the wrapper code too"""
    _, _, _, rfuncs, src = parse_rust()
    body = re.sub(r'extern "C" \{.*?\n\}', "", src, flags=re.S)
    unused = [n for n in rfuncs if not re.search(r"\b%s\(" % n, body)]
    assert not unused, unused


def test_balanced_delimiters():
    src = parse_rust()[4]
    src = re.sub(r'"(?:[^"\\]|\\.)*"', '""', src)
    src = re.sub(r"'(?:[^'\\]|\\.)'", "' '", src)
    for a, b in ("()", "[]", "{}"):
        assert src.count(a) == src.count(b), (a, src.count(a), src.count(b))


def test_type_mapping_examples():
    """the checker itself: a few C spellings and the Rust type each must be declared as"""
    enums, known = {"rp_sample_format"}, {"rp_ctx", "rp_templates"}
    for c, r in (("const float *const *", "*const *const f32"), ("rp_ctx *const *", "*const *mut rp_ctx"), ("uint8_t **", "*mut *mut u8"),
                 ("const rp_templates *const *", "*const *const rp_templates"), ("rp_sample_format", "c_int"), ("const void *const *", "*const *const c_void"),
                 ("long long", "i64"), ("const char *const *", "*const *const c_char"), ("void *", "*mut c_void"), ("const size_t *", "*const usize")):
        assert c_to_rust(c, enums, known) == r, (c, c_to_rust(c, enums, known), r)
