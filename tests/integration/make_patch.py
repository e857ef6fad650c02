#!/usr/bin/env python3
"""Regenerates tests/integration/smm_with_hip.patch: INTEGRATION.md section B as a real patch against the reference header
(vasil-pashov/sparse_matrix_math v0.2.0, include/sparse_matrix_math.h).  The hooks are pure INSERTIONS -- each `#if defined(SMM_WITH_HIP)`
block goes in front of the reference code it replaces -- so the zero-context unified diff written here holds only lines of this
repository (the `+` lines below and `@@` line numbers), never a line of the reference.  Build container only (needs /root/reference)."""
import hashlib
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference/include/sparse_matrix_math.h"
SHA256 = "2ac1d29caec1311c7d0800e16f128b6b65bcae1c4f39c7d61c430f02ea1f1c35"  # the header the line numbers below belong to

# (insert AFTER this line of the reference header, text).  Line numbers: ref:N in the comments of include/smm_hip.h.
HOOKS = [
    # ---- ref:1-14 the includes: the ABI and a few helpers --------------------------------------------------------------------
    (14, r'''
#if defined(SMM_WITH_HIP)
	#include <limits>
	#include <type_traits>
	#include "smm_hip.h"  // extern "C" ABI of libsmm_hip.so
namespace SMM {
	template<typename T> class CSRMatrix;
	namespace hip_detail {
		// status of the calling thread's last hot-path call: the reference's signatures have no room for it
		inline int& lastStatus() { static thread_local int st = SMM_HIP_OK; return st; }
		inline int note(int st) { lastStatus() = st; return st; }
		// move-only owner of an ABI handle (CSRMatrix and the preconditioners are move-only in the reference too)
		template<typename H, int (*Destroy)(H*)>
		struct Owned {
			mutable H* h = nullptr;
			Owned() noexcept = default;
			Owned(const Owned&) = delete;
			Owned& operator=(const Owned&) = delete;
			Owned(Owned&& o) noexcept : h(o.h) { o.h = nullptr; }
			Owned& operator=(Owned&& o) noexcept { reset(); h = o.h; o.h = nullptr; return *this; }
			~Owned() { reset(); }
			void reset() const noexcept { if (h) { Destroy(h); h = nullptr; } }
		};
		using Mirror = Owned<smm_hip_csr, smm_hip_csr_destroy>;
		using Precond = Owned<smm_hip_precond, smm_hip_precond_destroy>;
		inline int create(int r, int c, const int* s, const int* p, const float* v, smm_hip_csr** o) { return smm_hip_csr_create_f32(r, c, s, p, v, o); }
		inline int create(int r, int c, const int* s, const int* p, const double* v, smm_hip_csr** o) { return smm_hip_csr_create_f64(r, c, s, p, v, o); }
		inline int spmv(const smm_hip_csr* m, int op, const float* l, const float* x, float* o) { return smm_hip_spmv_f32(m, op, l, x, o); }
		inline int spmv(const smm_hip_csr* m, int op, const double* l, const double* x, double* o) { return smm_hip_spmv_f64(m, op, l, x, o); }
		inline int dot(int n, const float* a, const float* b, float* r) { return smm_hip_dot_f32(n, a, b, r); }
		inline int dot(int n, const double* a, const double* b, double* r) { return smm_hip_dot_f64(n, a, b, r); }
		inline int cg(const smm_hip_csr* a, const float* b, const float* x0, float* x, int it, float eps, int* st) { return smm_hip_cg_f32(a, b, x0, x, it, eps, nullptr, st, nullptr, nullptr); }
		inline int cg(const smm_hip_csr* a, const double* b, const double* x0, double* x, int it, double eps, int* st) { return smm_hip_cg_f64(a, b, x0, x, it, eps, nullptr, st, nullptr, nullptr); }
		inline int bicgstab(const smm_hip_csr* a, float* b, float* x, int it, float eps, const smm_hip_precond* M, int* st) { return smm_hip_bicgstab_f32(a, b, x, it, eps, M, st, nullptr, nullptr); }
		inline int bicgstab(const smm_hip_csr* a, double* b, double* x, int it, double eps, const smm_hip_precond* M, int* st) { return smm_hip_bicgstab_f64(a, b, x, it, eps, M, st, nullptr, nullptr); }
		inline int apply(const smm_hip_precond* M, const float* r, float* x) { return smm_hip_precond_apply_f32(M, r, x); }
		inline int apply(const smm_hip_precond* M, const double* r, double* x) { return smm_hip_precond_apply_f64(M, r, x); }
	}
	/// SMM_HIP_OK, or the SMM_HIP_ERR_* code of this thread's last call that went to the GPU
	inline int lastHipStatus() { return hip_detail::lastStatus(); }
}
#endif
'''),
    # ---- ref:305-306 Vector<T>::operator* : the dot product -------------------------------------------------------------------
    (306, r'''
#if defined(SMM_WITH_HIP)
		{
			T hipDot(0);
			if (hip_detail::note(hip_detail::dot(size, data, other.data, &hipDot)) != SMM_HIP_OK) hipDot = std::numeric_limits<T>::quiet_NaN();
			return hipDot;
		}
#endif
'''),
    # ---- ref:1173-1186 SGSPreconditioner: the device-side preconditioner, created on first use ------------------------------------
    (1185, r'''
#if defined(SMM_WITH_HIP)
			hip_detail::Precond hipM;
		public:
			const smm_hip_precond* hipHandle() const noexcept {
				if (!hipM.h && m.hip()) hip_detail::note(smm_hip_precond_create(m.hip(), SMM_PRECOND_SGS, &hipM.h));
				return hipM.h;
			}
		private:
#endif
'''),
    # ---- ref:1243-1259 CSRMatrix<T>: device mirror of the three arrays, next to them ----------------------------------------------
    (1259, r'''
#if defined(SMM_WITH_HIP)
		hip_detail::Mirror hipMirror;
	public:
		/// Device copy of values / positions / start, made on first use; nullptr (and lastHipStatus() != 0) without a GPU
		const smm_hip_csr* hip() const noexcept {
			if (!hipMirror.h && start) hip_detail::note(hip_detail::create(denseRowCount, denseColCount, start.get(), positions.get(), values.get(), &hipMirror.h));
			return hipMirror.h;
		}
		/// Every member that edits the arrays calls this (a CSRElement::setValue caller has to, too)
		void hipInvalidate() const noexcept { hipMirror.reset(); }
	private:
		void hipSpmv(int op, const T* lhs, const T* mult, T* out) const noexcept {
			if (!hip() || hip_detail::note(hip_detail::spmv(hip(), op, lhs, mult, out)) != SMM_HIP_OK) {
				for (int i = 0; i < denseRowCount; ++i) out[i] = std::numeric_limits<T>::quiet_NaN();  // never silently stale
			}
		}
#endif
'''),
    # ---- ref:1327 init, ref:1526-1597 the members that edit values: the mirror is stale ------------------------------------------
    (1327, "#if defined(SMM_WITH_HIP)\n\t\thipInvalidate();\n#endif\n"),
    (1526, "#if defined(SMM_WITH_HIP)\n\t\thipInvalidate();\n#endif\n"),
    (1534, "#if defined(SMM_WITH_HIP)\n\t\thipInvalidate();\n#endif\n"),
    (1543, "#if defined(SMM_WITH_HIP)\n\t\thipInvalidate();\n#endif\n"),
    (1573, "#if defined(SMM_WITH_HIP)\n\t\thipInvalidate();\n#endif\n"),
    (1592, "#if defined(SMM_WITH_HIP)\n\t\thipInvalidate();\n#endif\n"),
    (1597, "#if defined(SMM_WITH_HIP)\n\t\thipInvalidate();\n#endif\n"),
    # ---- ref:1501-1515 rMult / rMultAdd / rMultSub: the SpMV (in front of the call of rMultOp, ref:1458-1499) ---------------------
    (1503, "#if defined(SMM_WITH_HIP)\n\t\thipSpmv(SMM_OP_ASSIGN, nullptr, mult, res);\n\t\treturn;\n#endif\n"),
    (1508, "#if defined(SMM_WITH_HIP)\n\t\thipSpmv(SMM_OP_ADD, lhs, mult, out);\n\t\treturn;\n#endif\n"),
    (1513, "#if defined(SMM_WITH_HIP)\n\t\thipSpmv(SMM_OP_SUB, lhs, mult, out);\n\t\treturn;\n#endif\n"),
    # ---- ref:1658-1659 SGSPreconditioner::apply -----------------------------------------------------------------------------------
    (1659, r'''
#if defined(SMM_WITH_HIP)
		return hipHandle() && hip_detail::note(hip_detail::apply(hipHandle(), rhs, x)) == SMM_HIP_OK ? 0 : 1;
#endif
'''),
    # ---- ref:2191-2199 BiCGStab<Preconditioner, T>: the whole loop stays on the device for the reference's own preconditioner types ---
    (2199, r'''
#if defined(SMM_WITH_HIP)
		if constexpr (std::is_same<Preconditioner, typename CSRMatrix<T>::IDPreconditioner>::value ||
		              std::is_same<Preconditioner, typename CSRMatrix<T>::SGSPreconditioner>::value) {
			const smm_hip_precond* hipM = nullptr;
			if constexpr (std::is_same<Preconditioner, typename CSRMatrix<T>::SGSPreconditioner>::value) {
				hipM = preconditioner.hipHandle();
				if (!hipM) return SolverStatus::DIVERGED;  // lastHipStatus() says why
			}
			int hipSolverStatus = 0;
			if (!a.hip() || hip_detail::note(hip_detail::bicgstab(a.hip(), b, x, maxIterations, eps, hipM, &hipSolverStatus)) != SMM_HIP_OK) {
				return SolverStatus::DIVERGED;
			}
			return static_cast<SolverStatus>(hipSolverStatus);
		}
#endif
'''),
    # ---- ref:2316-2324 ConjugateGradient<T> ---------------------------------------------------------------------------------------
    (2324, r'''
#if defined(SMM_WITH_HIP)
		{
			int hipSolverStatus = 0;
			if (!a.hip() || hip_detail::note(hip_detail::cg(a.hip(), b, x0, x, maxIterations, eps, &hipSolverStatus)) != SMM_HIP_OK) {
				return SolverStatus::DIVERGED;  // lastHipStatus() says why
			}
			return static_cast<SolverStatus>(hipSolverStatus);
		}
#endif
'''),
]


def patched_text(original_lines):
    by_line = {}
    for after, text in HOOKS:
        by_line.setdefault(after, []).append(text.lstrip("\n"))
    out = []
    for n, line in enumerate(original_lines, start=1):
        out.append(line)
        for text in by_line.get(n, []):
            out.append(text if text.endswith("\n") else text + "\n")
    return "".join(out)


def main():
    raw = open(REFERENCE, "rb").read()
    if hashlib.sha256(raw).hexdigest() != SHA256:
        raise SystemExit("the mounted reference header is not the v0.2.0 file these line numbers belong to")
    lines = raw.decode("utf-8").splitlines(keepends=True)
    with tempfile.TemporaryDirectory() as tmp:
        a = os.path.join(tmp, "a", "sparse_matrix_math.h")
        b = os.path.join(tmp, "b", "sparse_matrix_math.h")
        os.makedirs(os.path.dirname(a))
        os.makedirs(os.path.dirname(b))
        open(a, "w", encoding="utf-8").write("".join(lines))
        open(b, "w", encoding="utf-8").write(patched_text(lines))
        r = subprocess.run(["diff", "-U0", "--label", "a/include/sparse_matrix_math.h", "--label", "b/include/sparse_matrix_math.h", a, b], capture_output=True, text=True)
    body = r.stdout
    assert r.returncode == 1 and body
    for ln in body.splitlines():
        assert ln.startswith(("+", "@@", "---")), f"a line of the reference would enter the patch: {ln!r}"
    header = ("# INTEGRATION.md section B as a patch: apply to include/sparse_matrix_math.h of vasil-pashov/sparse_matrix_math v0.2.0\n"
              f"# (sha256 {SHA256}) with   patch -p1 < smm_with_hip.patch   and build with -DSMM_WITH_HIP -I<repo>/include -lsmm_hip.\n"
              "# Insertions only (zero-context diff): every line below is this repository's; generated by tests/integration/make_patch.py.\n")
    open(os.path.join(HERE, "smm_with_hip.patch"), "w", encoding="utf-8").write(header + body)
    print(f"wrote smm_with_hip.patch: {sum(1 for ln in body.splitlines() if ln.startswith('+') and not ln.startswith('+++'))} inserted lines in {body.count('@@ -')} hunks")


if __name__ == "__main__":
    sys.exit(main())
