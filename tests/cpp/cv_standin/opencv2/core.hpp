// TEST-ONLY stand-in for the handful of OpenCV core types the drop-in header
// (include/slam_amd_opencv.hpp) and the reference's frame / map headers touch.  It exists so that
// the header can be compiled -- and driven end to end on the GPU -- where OpenCV is not
// installed.  It pins nothing about OpenCV's arithmetic: it has none.  Names, member names and
// record layouts follow the public OpenCV 4.x API (cv::Mat as a ref-counted row-major array,
// cv::KeyPoint 28 bytes, cv::DMatch 16 bytes).
#ifndef SLAM_AMD_TEST_CV_STANDIN_CORE_HPP
#define SLAM_AMD_TEST_CV_STANDIN_CORE_HPP

#include <cassert>
#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <vector>

#define CV_8U 0
#define CV_32F 5
#define CV_64F 6
#define CV_8UC1 CV_8U
#define CV_8UC3 16
#define CV_Assert(expr) do { if (!(expr)) throw std::runtime_error("CV_Assert: " #expr); } while (0)

namespace cv {

template <class T> using Ptr = std::shared_ptr<T>;
template <class T, class... A> Ptr<T> makePtr(A&&... a) { return std::make_shared<T>(std::forward<A>(a)...); }

struct Point2f { float x = 0, y = 0; Point2f() {} Point2f(float x, float y) : x(x), y(y) {} };
struct Point3f {
    float x = 0, y = 0, z = 0;
    Point3f() {}
    Point3f(float x, float y, float z) : x(x), y(y), z(z) {}
};

struct Point { int x = 0, y = 0; Point() {} Point(int x, int y) : x(x), y(y) {} };
struct Size {
    int width = 0, height = 0;
    Size() {}
    Size(int w, int h) : width(w), height(h) {}
    bool operator==(const Size& o) const { return width == o.width && height == o.height; }
};
struct Rect {
    int x = 0, y = 0, width = 0, height = 0;
    Rect() {}
    Rect(int x, int y, int w, int h) : x(x), y(y), width(w), height(h) {}
};
Rect operator&(const Rect& a, const Rect& b);   // declared only
struct Scalar { double val[4] = {0, 0, 0, 0}; Scalar() {} Scalar(double v0) { val[0] = v0; } };

struct KeyPoint {
    Point2f pt;
    float size = 0, angle = -1, response = 0;
    int octave = 0, class_id = -1;
};

struct DMatch {
    int queryIdx = -1, trainIdx = -1, imgIdx = -1;
    float distance = 0;
    DMatch() {}
    DMatch(int q, int t, int i, float d) : queryIdx(q), trainIdx(t), imgIdx(i), distance(d) {}
    bool operator<(const DMatch& m) const { return distance < m.distance; }
};

class Mat {
public:
    int rows = 0, cols = 0;
    size_t step = 0;
    uint8_t* data = nullptr;
    Mat() {}
    Mat(int r, int c, int type) { create(r, c, type); }
    Mat(int r, int c, int type, const Scalar& fill);   // declared only
    Size size() const { return Size(cols, rows); }
    uint8_t* ptr(int r = 0) { return data + static_cast<size_t>(r) * step; }
    const uint8_t* ptr(int r = 0) const { return data + static_cast<size_t>(r) * step; }
    static size_t elem(int type) { return type == CV_8U ? 1 : type == CV_32F ? 4 : type == CV_64F ? 8 : type == CV_8UC3 ? 3 : 0; }
    void create(int r, int c, int type) {
        type_ = type; rows = r; cols = c; step = static_cast<size_t>(c) * elem(type);
        store_ = std::make_shared<std::vector<uint8_t>>(static_cast<size_t>(r) * step);
        data = store_->empty() ? nullptr : store_->data();
    }
    bool empty() const { return data == nullptr || rows == 0 || cols == 0; }
    int type() const { return type_; }
    size_t total() const { return static_cast<size_t>(rows) * static_cast<size_t>(cols); }
    template <class T> T* ptr(int r = 0) { return reinterpret_cast<T*>(data + static_cast<size_t>(r) * step); }
    template <class T> const T* ptr(int r = 0) const { return reinterpret_cast<const T*>(data + static_cast<size_t>(r) * step); }
    template <class T> T& at(int r, int c) { return ptr<T>(r)[c]; }
    template <class T> const T& at(int r, int c) const { return ptr<T>(r)[c]; }
    template <class T> T& at(int i) { return ptr<T>(cols ? i / cols : 0)[cols ? i % cols : 0]; }
    template <class T> const T& at(int i) const { return ptr<T>(cols ? i / cols : 0)[cols ? i % cols : 0]; }
    Mat row(int r) const { Mat m = *this; m.rows = 1; m.data = data + static_cast<size_t>(r) * step; return m; }
    void convertTo(Mat& dst, int type) const {
        Mat out(rows, cols, type);
        for (int r = 0; r < rows; r++)
            for (int c = 0; c < cols; c++) {
                double v = type_ == CV_64F ? at<double>(r, c) : type_ == CV_32F ? static_cast<double>(at<float>(r, c))
                                                                                : static_cast<double>(at<uint8_t>(r, c));
                if (type == CV_64F) out.at<double>(r, c) = v;
                else if (type == CV_32F) out.at<float>(r, c) = static_cast<float>(v);
                else out.at<uint8_t>(r, c) = static_cast<uint8_t>(v);
            }
        dst = out;
    }
private:
    int type_ = CV_8U;
    std::shared_ptr<std::vector<uint8_t>> store_;
};

// ---- declarations without definitions: enough for a syntax check of code written against OpenCV, never linked ----
class _InputArray { public: _InputArray(); _InputArray(const Mat&); };
class _OutputArray : public _InputArray { public: _OutputArray(); _OutputArray(Mat&); };
class _InputOutputArray : public _OutputArray { public: _InputOutputArray(); _InputOutputArray(Mat&); };
typedef const _InputArray& InputArray;
typedef const _OutputArray& OutputArray;
typedef const _InputOutputArray& InputOutputArray;
InputOutputArray noArray();
enum NormTypes { NORM_L2 = 4, NORM_HAMMING = 6 };
void randn(InputOutputArray dst, const Scalar& mean, const Scalar& stddev);
void add(InputArray a, InputArray b, OutputArray dst, InputArray mask, int dtype = -1);

}  // namespace cv

#endif
