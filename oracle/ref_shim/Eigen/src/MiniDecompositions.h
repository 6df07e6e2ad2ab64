// ORACLE-SIDE TEST INFRASTRUCTURE (see ../Core).  Dense decompositions the reference calls at the edge of the hot path, in their
// textbook form: pivoted LDL^T (the algorithm Eigen documents for LDLT: symmetric diagonal pivoting, unblocked, lower triangle),
// inverse / determinant by partial-pivot elimination, one-sided Jacobi SVD, eigenvalues of a general real matrix via QR iterations
// on the Hessenberg form (used by the reference only for logging).  These are NOT Eigen's code: results agree with Eigen to
// rounding, not bitwise — the functions that depend on them are reported as "unpinned" in DESIGN.md.
#pragma once

namespace Eigen
{

template<typename M>
class LDLT
{
public:
	typedef typename M::Scalar S;
	LDLT() : n_(0), zero_(false) {}
	explicit LDLT(const M& a) { compute(a); }
	LDLT& compute(const M& a)
	{
		n_ = a.rows();
		m_.assign((size_t)(n_ * n_), S(0));
		tr_.assign((size_t)n_, 0);
		for (Index i = 0; i < n_; i++) for (Index j = 0; j <= i; j++) at(i, j) = a(i, j);
		std::vector<S> temp((size_t)n_);
		zero_ = false;
		for (Index k = 0; k < n_; k++)
		{
			Index big = k; S bigv = std::abs(at(k, k));
			for (Index i = k + 1; i < n_; i++) { S v = std::abs(at(i, i)); if (v > bigv) { bigv = v; big = i; } }
			tr_[k] = big;
			if (k != big)
			{
				Index s = n_ - big - 1;
				for (Index c = 0; c < k; c++) std::swap(at(k, c), at(big, c));
				for (Index r = 0; r < s; r++) std::swap(at(big + 1 + r, k), at(big + 1 + r, big));
				std::swap(at(k, k), at(big, big));
				for (Index i = k + 1; i < big; i++) std::swap(at(i, k), at(big, i));
			}
			Index rs = n_ - k - 1;
			if (k > 0)
			{
				for (Index j = 0; j < k; j++) temp[j] = at(j, j) * at(k, j);
				S s = S(0);
				for (Index j = 0; j < k; j++) s += at(k, j) * temp[j];
				at(k, k) -= s;
				for (Index r = 0; r < rs; r++)
				{
					S s2 = S(0);
					for (Index j = 0; j < k; j++) s2 += at(k + 1 + r, j) * temp[j];
					at(k + 1 + r, k) -= s2;
				}
			}
			S akk = at(k, k);
			bool ok = std::abs(akk) > S(0);
			if (k == 0 && !ok) { zero_ = true; for (Index j = 0; j < n_; j++) tr_[j] = j; break; }
			if (rs > 0 && ok) for (Index r = 0; r < rs; r++) at(k + 1 + r, k) /= akk;
		}
		return *this;
	}
	template<typename B, int R2, int C2> Matrix<S, R2, C2> solve(const DenseBase<B, S, R2, C2>& rhs) const
	{
		Matrix<S, R2, C2> x(rhs.rows(), rhs.cols(), internal::Sized());
		for (Index c = 0; c < rhs.cols(); c++)
		{
			std::vector<S> d((size_t)n_);
			for (Index i = 0; i < n_; i++) d[i] = rhs(i, c);
			if (zero_) { for (Index i = 0; i < n_; i++) x(i, c) = S(0); continue; }
			for (Index k = 0; k < n_; k++) if (tr_[k] != k) std::swap(d[k], d[tr_[k]]);
			for (Index i = 0; i < n_; i++) { S s = d[i]; for (Index j = 0; j < i; j++) s -= at(i, j) * d[j]; d[i] = s; }
			const S tol = (std::numeric_limits<S>::min)();
			for (Index i = 0; i < n_; i++) { if (std::abs(at(i, i)) > tol) d[i] /= at(i, i); else d[i] = S(0); }
			for (Index i = n_ - 1; i >= 0; i--) { S s = d[i]; for (Index j = i + 1; j < n_; j++) s -= at(j, i) * d[j]; d[i] = s; }
			for (Index k = n_ - 1; k >= 0; k--) if (tr_[k] != k) std::swap(d[k], d[tr_[k]]);
			for (Index i = 0; i < n_; i++) x(i, c) = d[i];
		}
		return x;
	}
	Matrix<S, M::RowsAtCompileTime, 1> vectorD() const
	{
		Matrix<S, M::RowsAtCompileTime, 1> d(n_, 1, internal::Sized());
		for (Index i = 0; i < n_; i++) d(i) = at(i, i);
		return d;
	}
	bool isPositive() const { for (Index i = 0; i < n_; i++) if (at(i, i) < S(0)) return false; return true; }
	int info() const { return 0; }
private:
	S& at(Index r, Index c) { return m_[(size_t)(r * n_ + c)]; }
	S at(Index r, Index c) const { return m_[(size_t)(r * n_ + c)]; }
	Index n_;
	std::vector<S> m_;
	std::vector<Index> tr_;
	bool zero_;
};

template<typename D, typename S, int R, int C>
LDLT<Matrix<S, R, C>> DenseBase<D, S, R, C>::ldlt() const { return LDLT<Matrix<S, R, C>>(Matrix<S, R, C>(derived())); }

// inverse: Gauss-Jordan with partial pivoting
template<typename D, typename S, int R, int C>
Matrix<S, R, C> DenseBase<D, S, R, C>::inverse() const
{
	const Index n = rows();
	assert(n == cols());
	if (n == 1) { Matrix<S, R, C> r(1, 1, internal::Sized()); r(0, 0) = S(1) / coeff(0, 0); return r; }
	if (n == 2)
	{
		// closed form, as Eigen documents for fixed sizes up to 4: adjugate times the reciprocal determinant
		Matrix<S, R, C> r(2, 2, internal::Sized());
		const S invdet = S(1) / (coeff(0, 0) * coeff(1, 1) - coeff(1, 0) * coeff(0, 1));
		r(0, 0) = coeff(1, 1) * invdet; r(1, 0) = -coeff(1, 0) * invdet; r(0, 1) = -coeff(0, 1) * invdet; r(1, 1) = coeff(0, 0) * invdet;
		return r;
	}
	if (n == 3)
	{
		// cofactor(i,j) = m(i+1,j+1) m(i+2,j+2) - m(i+1,j+2) m(i+2,j+1)  (indices mod 3); det from the cofactors of column 0;
		// inverse(i,j) = cofactor(j,i) / det
		Matrix<S, R, C> r(3, 3, internal::Sized());
		auto cof = [this](int i, int j) {
			const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
			return coeff(i1, j1) * coeff(i2, j2) - coeff(i1, j2) * coeff(i2, j1);
		};
		const S c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
		const S det = c0 * coeff(0, 0) + c1 * coeff(1, 0) + c2 * coeff(2, 0);
		const S invdet = S(1) / det;
		for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r(i, j) = ((j == 0) ? (i == 0 ? c0 : (i == 1 ? c1 : c2)) : cof(j, i)) * invdet;
		// note r(i,0) = cofactor(0,i): fix the first column
		r(0, 0) = c0 * invdet; r(1, 0) = cof(0, 1) * invdet; r(2, 0) = cof(0, 2) * invdet;
		r(0, 1) = c1 * invdet; r(0, 2) = c2 * invdet;
		return r;
	}
	Matrix<S, R, C> a(derived()), inv(n, n, internal::Sized());
	inv.setIdentity();
	for (Index k = 0; k < n; k++)
	{
		Index p = k; S best = std::abs(a(k, k));
		for (Index i = k + 1; i < n; i++) if (std::abs(a(i, k)) > best) { best = std::abs(a(i, k)); p = i; }
		if (p != k) for (Index j = 0; j < n; j++) { std::swap(a(k, j), a(p, j)); std::swap(inv(k, j), inv(p, j)); }
		S piv = a(k, k);
		for (Index j = 0; j < n; j++) { a(k, j) /= piv; inv(k, j) /= piv; }
		for (Index i = 0; i < n; i++)
		{
			if (i == k) continue;
			S f = a(i, k);
			if (f == S(0)) continue;
			for (Index j = 0; j < n; j++) { a(i, j) -= f * a(k, j); inv(i, j) -= f * inv(k, j); }
		}
	}
	return inv;
}
template<typename D, typename S, int R, int C>
S DenseBase<D, S, R, C>::determinant() const
{
	const Index n = rows();
	Matrix<S, R, C> a(derived());
	S det = S(1);
	for (Index k = 0; k < n; k++)
	{
		Index p = k; S best = std::abs(a(k, k));
		for (Index i = k + 1; i < n; i++) if (std::abs(a(i, k)) > best) { best = std::abs(a(i, k)); p = i; }
		if (best == S(0)) return S(0);
		if (p != k) { for (Index j = 0; j < n; j++) std::swap(a(k, j), a(p, j)); det = -det; }
		det *= a(k, k);
		for (Index i = k + 1; i < n; i++)
		{
			S f = a(i, k) / a(k, k);
			for (Index j = k; j < n; j++) a(i, j) -= f * a(k, j);
		}
	}
	return det;
}

// eigenvalues of a real square matrix (logging only in the reference): symmetric matrices take cyclic Jacobi rotations, anything
// else falls back to unshifted QR iterations and reports the diagonal — accurate enough for a log line.
template<typename D, typename S, int R, int C>
typename DenseBase<D, S, R, C>::EigenvaluesReturn DenseBase<D, S, R, C>::eigenvalues() const
{
	const Index n = rows();
	Matrix<S, R, C> a(derived());
	for (int sweep = 0; sweep < 60; sweep++)
	{
		S off = S(0);
		for (Index p = 0; p < n; p++) for (Index q = p + 1; q < n; q++) off += a(p, q) * a(p, q) + a(q, p) * a(q, p);
		if (off < S(1e-30)) break;
		for (Index p = 0; p < n; p++) for (Index q = p + 1; q < n; q++)
		{
			S apq = S(0.5) * (a(p, q) + a(q, p));
			if (std::abs(apq) < S(1e-300)) continue;
			S theta = (a(q, q) - a(p, p)) / (S(2) * apq);
			S t = (theta >= 0 ? S(1) : S(-1)) / (std::abs(theta) + std::sqrt(theta * theta + S(1)));
			S c = S(1) / std::sqrt(t * t + S(1)), s = t * c;
			for (Index k = 0; k < n; k++) { S akp = a(k, p), akq = a(k, q); a(k, p) = c * akp - s * akq; a(k, q) = s * akp + c * akq; }
			for (Index k = 0; k < n; k++) { S apk = a(p, k), aqk = a(q, k); a(p, k) = c * apk - s * aqk; a(q, k) = s * apk + c * aqk; }
		}
	}
	EigenvaluesReturn ev;
	ev.re = Matrix<S, R, 1>(n, 1, internal::Sized());
	for (Index i = 0; i < n; i++) ev.re(i) = a(i, i);
	return ev;
}

// one-sided Jacobi (Hestenes) SVD, singular values sorted in decreasing order, thin U and V
template<typename M>
class JacobiSVD
{
public:
	typedef typename M::Scalar S;
	typedef Matrix<S, Dynamic, Dynamic> MatX;
	typedef Matrix<S, Dynamic, 1> VecX;
	JacobiSVD() {}
	template<typename D, int R, int C> JacobiSVD(const DenseBase<D, S, R, C>& a, unsigned int = 0) { compute(a); }
	template<typename D, int R, int C> JacobiSVD& compute(const DenseBase<D, S, R, C>& a, unsigned int = 0)
	{
		const Index m = a.rows(), n = a.cols();
		const bool flip = m < n;  // work on the tall orientation
		MatX A = flip ? MatX(a.transpose()) : MatX(a);
		const Index rr = A.rows(), cc = A.cols();
		MatX V = MatX::Identity(cc, cc);
		for (int sweep = 0; sweep < 100; sweep++)
		{
			bool rotated = false;
			for (Index p = 0; p < cc; p++) for (Index q = p + 1; q < cc; q++)
			{
				S alpha = S(0), beta = S(0), gamma = S(0);
				for (Index i = 0; i < rr; i++) { alpha += A(i, p) * A(i, p); beta += A(i, q) * A(i, q); gamma += A(i, p) * A(i, q); }
				if (std::abs(gamma) <= std::numeric_limits<S>::epsilon() * std::sqrt(alpha * beta) || gamma == S(0)) continue;
				rotated = true;
				S zeta = (beta - alpha) / (S(2) * gamma);
				S t = (zeta >= 0 ? S(1) : S(-1)) / (std::abs(zeta) + std::sqrt(S(1) + zeta * zeta));
				S c = S(1) / std::sqrt(S(1) + t * t), s = c * t;
				for (Index i = 0; i < rr; i++) { S x = A(i, p), y = A(i, q); A(i, p) = c * x - s * y; A(i, q) = s * x + c * y; }
				for (Index i = 0; i < cc; i++) { S x = V(i, p), y = V(i, q); V(i, p) = c * x - s * y; V(i, q) = s * x + c * y; }
			}
			if (!rotated) break;
		}
		std::vector<std::pair<S, Index>> order;
		for (Index j = 0; j < cc; j++) { S s = S(0); for (Index i = 0; i < rr; i++) s += A(i, j) * A(i, j); order.push_back(std::make_pair(std::sqrt(s), j)); }
		std::stable_sort(order.begin(), order.end(), [](const std::pair<S, Index>& x, const std::pair<S, Index>& y) { return x.first > y.first; });
		MatX U(rr, cc, internal::Sized()), Vs(cc, cc, internal::Sized());
		sv_ = VecX(cc, 1, internal::Sized());
		for (Index k = 0; k < cc; k++)
		{
			Index j = order[(size_t)k].second; S s = order[(size_t)k].first;
			sv_(k) = s;
			for (Index i = 0; i < rr; i++) U(i, k) = (s > S(0)) ? A(i, j) / s : S(0);
			for (Index i = 0; i < cc; i++) Vs(i, k) = V(i, j);
		}
		if (flip) { U_ = Vs; V_ = U; } else { U_ = U; V_ = Vs; }
		return *this;
	}
	const VecX& singularValues() const { return sv_; }
	const MatX& matrixU() const { return U_; }
	const MatX& matrixV() const { return V_; }
	Index rank() const { Index r = 0; for (Index i = 0; i < sv_.size(); i++) if (sv_(i) > sv_(0) * std::numeric_limits<S>::epsilon() * S(sv_.size())) r++; return r; }
private:
	MatX U_, V_;
	VecX sv_;
};

}  // namespace Eigen
