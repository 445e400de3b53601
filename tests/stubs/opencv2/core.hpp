// TESTS-ONLY minimal stand-in for <opencv2/core.hpp>.  NOT OpenCV, not a reference build, not parity evidence: it exists so
// that the cv::Mat-typed constructor of the shim's SemanticPlane (`#ifdef MLD_HAVE_OPENCV`; reference RansacPlane.h:189)
// is compiled and called at least once in this image, which has no OpenCV.  An 8-bit matrix with rows / cols / step /
// data, type(), ptr<T>(row), at<T>(row, col) - the members the shim and its call site touch.
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>
#include <vector>

#define CV_8U 0
#define CV_8UC1 0
#define CV_8UC3 16

namespace cv {

class Mat {
public:
    Mat() = default;
    Mat(int rows_, int cols_, int type_) : rows(rows_), cols(cols_), step((size_t)cols_ * (size_t)(1 + (type_ >> 3))), _type(type_) {
        _buf = std::make_shared<std::vector<std::uint8_t>>((size_t)rows_ * step);
        data = _buf->data();
    }
    // a view of caller memory (cv::Mat(rows, cols, type, data, step))
    Mat(int rows_, int cols_, int type_, void* data_, size_t step_) : rows(rows_), cols(cols_), step(step_), _type(type_) {
        data = static_cast<std::uint8_t*>(data_);
    }
    int type() const { return _type; }
    int channels() const { return 1 + (_type >> 3); }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    template <typename T>
    T* ptr(int r) {
        return reinterpret_cast<T*>(data + (size_t)r * step);
    }
    template <typename T>
    const T* ptr(int r) const {
        return reinterpret_cast<const T*>(data + (size_t)r * step);
    }
    template <typename T>
    T& at(int r, int c) {
        return ptr<T>(r)[c];
    }
    template <typename T>
    const T& at(int r, int c) const {
        return ptr<T>(r)[c];
    }
    int rows = 0, cols = 0;
    size_t step = 0;
    std::uint8_t* data = nullptr;

private:
    int _type = CV_8UC1;
    std::shared_ptr<std::vector<std::uint8_t>> _buf;
};

}  // namespace cv
