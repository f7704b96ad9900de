// mini_eigen_sparse.hpp -- stand-in for the part of Eigen::SparseMatrix<double> (column major) that minisam's linearisation
// (vendor/minisam/minisam/nonlinear/linearization.cpp) drives: construction, reserve(per-column sizes) -> uncompressed storage
// whose index arrays the caller overwrites through the raw pointers, insert(i, j), makeCompressed(), selfadjointView<Lower>()
// -> full symmetric matrix.  Written from scratch for the checker build (oracle/Makefile.ref); layout follows Eigen 3.3's
// published storage scheme (outerIndex / innerNonZeros / innerIndex / values).  Test infrastructure only.
#pragma once
#include <vector>

namespace Eigen {

template <class Scalar_, int Options_ = 0, class StorageIndex_ = int>
class SparseMatrix {
public:
    typedef Scalar_ Scalar;
    typedef StorageIndex_ StorageIndex;
    SparseMatrix() : rows_(0), cols_(0), compressed_(true) { outer_.assign(1, 0); }
    SparseMatrix(Index rows, Index cols) : rows_(rows), cols_(cols), compressed_(true) { outer_.assign((size_t)cols + 1, 0); }

    Index rows() const { return rows_; }
    Index cols() const { return cols_; }
    Index outerSize() const { return cols_; }
    bool isCompressed() const { return compressed_; }
    Index nonZeros() const
    {
        if (compressed_) return outer_[(size_t)cols_];
        Index n = 0; for (Index j = 0; j < cols_; ++j) n += nnz_[(size_t)j]; return n;
    }
    // reserve room for sizes[j] entries in column j of an EMPTY matrix: the matrix becomes uncompressed, column j starts at the
    // running sum of the reserved sizes
    template <class SizesType> void reserve(const SizesType& sizes)
    {
        compressed_ = false;
        nnz_.assign((size_t)cols_, 0);
        StorageIndex total = 0;
        for (Index j = 0; j < cols_; ++j) { outer_[(size_t)j] = total; total += (StorageIndex)sizes[(size_t)j]; }
        outer_[(size_t)cols_] = total;
        values_.assign((size_t)total, Scalar(0));
        inner_.assign((size_t)total, 0);
    }
    Scalar* valuePtr() { return values_.data(); }
    const Scalar* valuePtr() const { return values_.data(); }
    StorageIndex* innerIndexPtr() { return inner_.data(); }
    const StorageIndex* innerIndexPtr() const { return inner_.data(); }
    StorageIndex* outerIndexPtr() { return outer_.data(); }
    const StorageIndex* outerIndexPtr() const { return outer_.data(); }
    StorageIndex* innerNonZeroPtr() { return compressed_ ? nullptr : nnz_.data(); }
    const StorageIndex* innerNonZeroPtr() const { return compressed_ ? nullptr : nnz_.data(); }

    // a new entry (i, j) in reserved space; entries of a column must arrive with increasing i or are sorted into place
    Scalar& insert(Index i, Index j)
    {
        if (compressed_) { std::vector<StorageIndex> one((size_t)cols_, 8); reserveKeeping_(one); }
        const StorageIndex start = outer_[(size_t)j];
        StorageIndex p = start + nnz_[(size_t)j];
        if (p >= outer_[(size_t)j + 1]) { std::vector<StorageIndex> more((size_t)cols_, 0); more[(size_t)j] = nnz_[(size_t)j] + 8; reserveKeeping_(more); return insert(i, j); }
        while (p > start && inner_[(size_t)p - 1] > (StorageIndex)i) { inner_[(size_t)p] = inner_[(size_t)p - 1]; values_[(size_t)p] = values_[(size_t)p - 1]; --p; }
        inner_[(size_t)p] = (StorageIndex)i; values_[(size_t)p] = Scalar(0);
        ++nnz_[(size_t)j];
        return values_[(size_t)p];
    }
    void makeCompressed()
    {
        if (compressed_) return;
        StorageIndex dst = 0;
        for (Index j = 0; j < cols_; ++j) {
            const StorageIndex src = outer_[(size_t)j], n = nnz_[(size_t)j];
            if (src != dst) for (StorageIndex k = 0; k < n; ++k) { inner_[(size_t)(dst + k)] = inner_[(size_t)(src + k)]; values_[(size_t)(dst + k)] = values_[(size_t)(src + k)]; }
            outer_[(size_t)j] = dst;
            dst += n;
        }
        outer_[(size_t)cols_] = dst;
        values_.resize((size_t)dst); inner_.resize((size_t)dst);
        nnz_.clear();
        compressed_ = true;
    }
    Scalar coeff(Index i, Index j) const
    {
        const StorageIndex a = outer_[(size_t)j], b = compressed_ ? outer_[(size_t)j + 1] : a + nnz_[(size_t)j];
        for (StorageIndex p = a; p < b; ++p) if (inner_[(size_t)p] == (StorageIndex)i) return values_[(size_t)p];
        return Scalar(0);
    }
    template <int UpLo> struct SelfAdjointViewS { const SparseMatrix* m; };
    template <int UpLo> SelfAdjointViewS<UpLo> selfadjointView() const { return SelfAdjointViewS<UpLo>{this}; }
    // full symmetric matrix from the stored triangle
    template <int UpLo> SparseMatrix& operator=(const SelfAdjointViewS<UpLo>& v)
    {
        const SparseMatrix& a = *v.m;
        SparseMatrix r(a.rows_, a.cols_);
        std::vector<StorageIndex> cnt((size_t)a.cols_, 0);
        auto each = [&](auto&& f) {
            for (Index j = 0; j < a.cols_; ++j) {
                const StorageIndex s0 = a.outer_[(size_t)j], s1 = a.compressed_ ? a.outer_[(size_t)j + 1] : s0 + a.nnz_[(size_t)j];
                for (StorageIndex p = s0; p < s1; ++p) {
                    const Index i = a.inner_[(size_t)p];
                    if ((UpLo == Lower) ? (i < j) : (i > j)) continue;
                    f(i, j, a.values_[(size_t)p]);
                    if (i != j) f(j, i, a.values_[(size_t)p]);
                }
            }
        };
        each([&](Index, Index j, Scalar) { ++cnt[(size_t)j]; });
        r.reserve(cnt);
        each([&](Index i, Index j, Scalar val) { r.insert(i, j) = val; });
        r.makeCompressed();
        *this = r;
        return *this;
    }

private:
    template <class SizesType> void reserveKeeping_(const SizesType& extra)      // grow the per-column room, keeping the entries
    {
        SparseMatrix old = *this;
        const bool was = compressed_;
        std::vector<StorageIndex> sizes((size_t)cols_);
        for (Index j = 0; j < cols_; ++j) {
            const StorageIndex have = was ? old.outer_[(size_t)j + 1] - old.outer_[(size_t)j] : old.nnz_[(size_t)j];
            sizes[(size_t)j] = have + (StorageIndex)extra[(size_t)j];
        }
        reserve(sizes);
        for (Index j = 0; j < cols_; ++j) {
            const StorageIndex have = was ? old.outer_[(size_t)j + 1] - old.outer_[(size_t)j] : old.nnz_[(size_t)j];
            for (StorageIndex k = 0; k < have; ++k) { inner_[(size_t)(outer_[(size_t)j] + k)] = old.inner_[(size_t)(old.outer_[(size_t)j] + k)]; values_[(size_t)(outer_[(size_t)j] + k)] = old.values_[(size_t)(old.outer_[(size_t)j] + k)]; }
            nnz_[(size_t)j] = have;
        }
    }
    Index rows_, cols_;
    bool compressed_;
    std::vector<Scalar> values_;
    std::vector<StorageIndex> inner_, outer_, nnz_;
};

} // namespace Eigen
