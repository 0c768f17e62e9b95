"""Properties of the source tree the task contract states and a reviewer greps for (CPU, no GPU, no library calls):
  * the product never imports, loads or executes anything under oracle/ (the oracle is test infrastructure);
  * the library reads no environment variable outside tool builds;
  * no compatibility layers: no hipify artefacts, CUDA headers, platform #ifdefs, Triton, or vendor GEMM / conv libraries behind the C ABI;
  * every preprocessor conditional in csrc/ is a documented tool-build switch;
  * the built library links only the HIP runtime and system libraries."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

PKG = os.path.join(ROOT, "livespeechportraits_amd")
CSRC = os.path.join(PKG, "csrc")


def _files(top, exts):
    for d, _, names in os.walk(top):
        if os.path.basename(d) in ("build", "__pycache__"):
            continue
        for n in names:
            if n.endswith(exts):
                yield os.path.join(d, n)


def _code_lines(path):
    """(line number, text) of a Python file without comments and docstring bodies (a mention of oracle/ in prose is fine)"""
    import io
    import tokenize
    src = open(path).read()
    keep = {}
    for tok in tokenize.generate_tokens(io.StringIO(src).readline):
        if tok.type in (tokenize.COMMENT, tokenize.STRING, tokenize.NL, tokenize.NEWLINE, tokenize.INDENT, tokenize.DEDENT):
            continue
        keep.setdefault(tok.start[0], []).append(tok.string)
    return [(n, " ".join(v)) for n, v in sorted(keep.items())]


def test_the_product_never_touches_the_oracle():
    bad = []
    for path in list(_files(PKG, (".py",))) + list(_files(os.path.join(ROOT, "integration"), (".py",))):
        for n, text in _code_lines(path):
            if re.search(r"\boracle\b", text):
                bad.append("%s:%d: %s" % (os.path.relpath(path, ROOT), n, text))
        # string literals could still carry a path to dlopen / exec: none may name the oracle directory
        for m in re.finditer(r"""["'][^"'\n]*oracle[/._][^"'\n]*["']""", open(path).read()):
            s = m.group(0)
            if re.search(r"oracle/[\w_]+\.(so|py|c)\b", s) and "see " not in s and len(s) < 80:
                bad.append("%s: literal %s" % (os.path.relpath(path, ROOT), s))
    assert not bad, bad
    for path in _files(CSRC, (".hip", ".cpp", ".h")):
        for n, line in enumerate(open(path), 1):
            code = line.split("//")[0]
            assert "oracle" not in code, "%s:%d" % (path, n)


def test_the_library_reads_no_environment_outside_tool_builds():
    for path in _files(CSRC, (".hip", ".cpp", ".h")):
        lines = open(path).read().split("\n")
        depth_tool = []                                           # stack of enclosing #if conditions
        for n, line in enumerate(lines, 1):
            s = line.strip()
            if s.startswith("#if"):
                depth_tool.append(s)
            elif s.startswith("#endif") and depth_tool:
                depth_tool.pop()
            code = line.split("//")[0]
            if re.search(r"\bgetenv\b|\benviron\b|secure_getenv", code):
                assert any("LSPF2F_ABLATE" in c for c in depth_tool), "%s:%d reads the environment in a product build: %s" % (os.path.relpath(path, ROOT), n, s)


ALLOWED_MACROS = re.compile(r"__HIPCC__|__cplusplus|LSPF2F_[A-Z0-9_]+|LSP[A-Z0-9]*_[A-Z0-9_]*H_?|WINO_ABL_[A-Z0-9_]+|W4_ABL_[A-Z0-9_]+|LC_[A-Z0-9_]+|FK_[A-Z0-9_]+|A2H_[A-Z0-9_]+|RASTER_[A-Z0-9_]+|[A-Z0-9_]+_STAMPS")


def test_no_compatibility_layers_and_only_tool_build_conditionals():
    banned = re.compile(r"__HIP_PLATFORM|__CUDA|cuda_runtime|cudaStream|hipify|triton|rocblas|hipblas|miopen|\bck_tile\b|composable_kernel|rocwmma", re.I)
    for path in list(_files(CSRC, (".hip", ".cpp", ".h"))) + list(_files(os.path.join(ROOT, "include"), (".h",))):
        for n, line in enumerate(open(path), 1):
            code = line.split("//")[0]
            assert not banned.search(code), "%s:%d: %s" % (os.path.relpath(path, ROOT), n, line.strip())
            s = code.strip()
            if s.startswith(("#ifdef", "#ifndef", "#if ", "#elif")):
                names = re.findall(r"[A-Za-z_][A-Za-z0-9_]*", re.sub(r"^#\s*\w+", "", s))
                for name in names:
                    if name in ("defined",):
                        continue
                    assert ALLOWED_MACROS.fullmatch(name), "%s:%d: conditional on %s (not a tool-build switch of this tree)" % (os.path.relpath(path, ROOT), n, name)
    for path in _files(PKG, (".py",)):
        for n, text in _code_lines(path):
            assert not re.search(r"\btriton\b|torch\.compile|\bjax\b", text), "%s:%d" % (path, n)


def test_the_built_library_links_only_hip_and_system_libraries():
    so = os.path.join(PKG, "liblspf2f.so")
    if not os.path.exists(so):
        pytest.skip("library not built (build() makes it)")
    out = subprocess.run(["ldd", so], capture_output=True, text=True).stdout
    libs = [l.split()[0] for l in out.strip().split("\n") if l.strip()]
    ok = re.compile(r"linux-vdso|libamdhip64|libstdc\+\+|libm\.|libgcc_s|libc\.|libpthread|libdl|librt|ld-linux|libnuma|libdrm|libelf|libz|libhsa|librocprofiler|libtinfo|libamd_comgr|libLLVM|libzstd|liblzma|libxml2|libicu|librocm")
    foreign = [l for l in libs if not ok.search(l)]
    assert not foreign, foreign
    assert not [l for l in libs if re.search(r"torch|rocblas|hipblas|MIOpen|rccl", l, re.I)], libs      # PyTorch and the vendor libraries are not behind the C ABI


def _declared_functions():
    """names of the C functions include/*.h declare (a declaration = `<type> name(` at the start of a line, outside comments and macros)"""
    names = set()
    for path in _files(os.path.join(ROOT, "include"), (".h",)):
        src = re.sub(r"/\*.*?\*/", "", open(path).read(), flags=re.S)
        for line in src.split("\n"):
            if line.startswith(("#", " ", "\t", "}", "typedef")):
                continue
            m = re.match(r"[A-Za-z_][A-Za-z0-9_ \*]*?\b(lsp[a-z0-9]+_[A-Za-z0-9_]+)\s*\(", line)
            if m:
                names.add(m.group(1))
    return names


def test_the_library_exports_its_headers_and_nothing_else():
    """A thin C-ABI library: `nm -D` lists exactly the functions include/*.h declares -- no mangled C++ internals, kernel stubs or template
    instantiations (csrc/Makefile: -fvisibility=hidden + csrc/exports.map; the headers open a `#pragma GCC visibility push(default)` region)."""
    so = os.path.join(PKG, "liblspf2f.so")
    if not os.path.exists(so):
        pytest.skip("library not built (build() makes it)")
    out = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    exported = {l.split()[-1] for l in out.strip().split("\n") if l.strip()}
    declared = _declared_functions()
    assert len(declared) > 60, sorted(declared)
    assert not (exported - declared), "exported but not declared in include/*.h: %s" % sorted(exported - declared)
    assert not (declared - exported), "declared in include/*.h but not exported: %s" % sorted(declared - exported)
    assert not [s for s in exported if s.startswith("_Z")]


def test_tune_keys_of_the_python_host_are_exactly_the_keys_the_library_parses():
    """LSP_HIP_<KEY> environment variables reach lspf2f_create_tuned only when <key> is in _native.TUNE_KEYS; a key the library gained and the list did not is warned
    about and DROPPED, so an A-B run silently measures the default (fullk16 / fullk16_min_frames).  The list must equal what api.cpp parses, and the header
    must name every key."""
    import re
    from livespeechportraits_amd import _native as N
    src = open(os.path.join(ROOT, "livespeechportraits_amd", "csrc", "api.cpp")).read()
    body = src[src.index("int lspf2f_create_tuned("):src.index("int lspf2f_destroy(")]
    parsed = set(re.findall(r'k == "([a-z0-9_]+)"', body))
    assert parsed, "no keys found: has the parser moved?"
    assert parsed == set(N.TUNE_KEYS), {"only in api.cpp": sorted(parsed - set(N.TUNE_KEYS)), "only in TUNE_KEYS": sorted(set(N.TUNE_KEYS) - parsed)}
    hdr = open(os.path.join(ROOT, "include", "lspf2f.h")).read()
    doc = hdr[hdr.index("Keys (default)"):hdr.index("int lspf2f_create_tuned(")]
    missing = [k for k in sorted(parsed) if not re.search(r"\b%s\b" % re.escape(k), doc)]
    assert not missing, "keys lspf2f_create_tuned parses that include/lspf2f.h does not document: %s" % missing
