// include/fits_output.h -- FITS image writer with the API of the reference's src/include/fits_output.h:46-358 and the
// byte layout cfitsio 3.47 gives the files that API produces -- without linking cfitsio.
//
// What an application can do with it (everything the reference's imageplane_disc_image / caustic_* programs use):
//   FITSOutput<double> fits(name);  fits.create_primary();                       // empty primary HDU
//   fits.write_image(T** data, Nx, Ny, transpose, flip_x, flip_y);               // BITPIX = -64 IMAGE extension
//   fits.write_image_array(double* frame, Nx, Ny);  fits.write_me_data_cube(...)
//   fits.set_ext_name(...), fits.write_comment(...), fits.write_keyword(name, comment, int|long|double|bool|string)
// Keywords go to the HDU created last, in call order, after the mandatory cards.  Binary tables (create_table /
// write_table_column, reference :209-262) are not implemented: none of the programs on this path writes one; the two
// entry points throw FITSOutputException.
//
// Byte format (the FITS standard plus the conventions cfitsio applies, checked byte for byte against files written by
// the cfitsio builds of the reference's applications, tests/test_host_formats.py):
//   * 2880-byte blocks; a header is 80-character cards, `END`, blank-padded; data big-endian, zero-padded.
//   * `KEYWORD = value / comment`: name upper-cased and blank-padded to 8, value right-aligned to column 30
//     (strings: quoted, blank-padded to 8 characters, left-aligned from column 11, quotes doubled, 68 max), comment
//     cut at column 80.  Doubles print with "%.15G" and always carry a decimal point ("30." , "1.93548387096774").
//   * a name longer than 8 characters, or with characters outside [A-Z0-9_-], becomes `HIERARCH NAME = value`.
//   * COMMENT cards carry 72 characters each; the primary HDU starts with cfitsio's two standard COMMENT lines.
//   * bytes outside printable ASCII (e.g. a UTF-8 dash in a comment) are written as blanks.
//
// Throughput: a 4096 x 4096 run writes six 134-MB images, and serially (transpose + byte swap into a zero-filled vector, then one ofstream
// write) that was 1.3 s of the program's 1.85 s on the GPU box.  An image is now transposed / byte-swapped by a team of threads into one of
// two reusable buffers and handed to a background writer that pwrite()s it in parallel slices while the application fills the next one
// (KRTRACE_HOST_THREADS caps the team, default min(hardware threads, 16)).  Same bytes in the file.
#ifndef FITS_OUTPUT_H_
#define FITS_OUTPUT_H_

#include <cctype>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <fstream>
#include <iostream>
#include <string>
#include <thread>
#include <vector>
#include <fcntl.h>
#include <unistd.h>
using namespace std;

#define COL_STRING "8A"
#define COL_STRING16 "16A"
#define COL_STRING32 "32A"
#define COL_STRING64 "64A"
#define COL_INT "1I"
#define COL_UINT "1U"
#define COL_INT32 "1J"
#define COL_UINT32 "1V"
#define COL_INT64 "1K"
#define COL_FLOAT "1E"
#define COL_FLOAT64 "1D"
#define COL_COMPLEX "1C"
#define COL_COMPLEX64 "1M"

class FITSOutputException : public exception {
public:
    explicit FITSOutputException(const string& msg, int status = 0) : text_("FITSOutput ERROR : " + msg)
    {
        if (status > 0) cerr << "FITSOutput: status " << status << endl;
    }
    const char* what() const noexcept override { return text_.c_str(); }

private:
    string text_;
};

namespace krhost {
namespace fits {

constexpr int kCard = 80, kBlock = 2880;

inline bool plain_keyword(const string& name)   // usable as a classic 8-character keyword?
{
    if (name.size() > 8) return false;
    for (char ch : name)
        if (!(isupper(static_cast<unsigned char>(ch)) || isdigit(static_cast<unsigned char>(ch)) || ch == '-' || ch == '_')) return false;
    return true;
}

inline string quoted(const string& text)         // 'text    ' : quotes doubled, at most 68 characters, at least 8
{
    string out = "'";
    for (size_t i = 0; i < text.size() && out.size() < 69; ++i) {
        out += text[i];
        if (text[i] == '\'' && out.size() < 69) out += '\'';
    }
    while (out.size() < 9) out += ' ';
    out += '\'';
    return out;
}

inline string real_text(double v)                // "%.15G" with a guaranteed decimal point
{
    char buf[64];
    snprintf(buf, sizeof buf, "%.15G", v);
    if (!strchr(buf, '.') && strchr(buf, 'E')) snprintf(buf, sizeof buf, "%.1E", v);
    if (strchr(buf, 'N')) throw FITSOutputException("keyword value is not a finite number");
    string s(buf);
    if (s.find('.') == string::npos && s.find('E') == string::npos) s += '.';
    return s;
}

// one `name = value / comment` card (value already formatted; quoted values start with an apostrophe)
inline string value_card(const string& raw_name, const string& value, const string& comment)
{
    string name = raw_name;
    name.erase(0, name.find_first_not_of(' '));
    if (!name.empty()) name.erase(name.find_last_not_of(' ') + 1);
    string card;
    if (plain_keyword(name)) {
        card = name;
        card.resize(8, ' ');
        card += "= ";
    } else {
        // decided on the name as given (a short lower-case name takes this branch too), then upper-cased like every keyword
        string upper = name;
        for (char& ch : upper) ch = static_cast<char>(toupper(static_cast<unsigned char>(ch)));
        card = "HIERARCH " + upper;
        card += (card.size() + 3 + value.size() > static_cast<size_t>(kCard)) ? "= " : " = ";
    }
    const size_t lead = card.size();
    if (!value.empty() && value[0] == '\'') {
        card += value.substr(0, kCard - lead);
        if (card.size() >= static_cast<size_t>(kCard)) card[kCard - 1] = '\'';
        if (card.size() < 30) card.resize(30, ' ');
    } else {
        if (lead + value.size() > static_cast<size_t>(kCard)) throw FITSOutputException("keyword value too long: " + name);
        if (lead + value.size() < 30) card.append(30 - lead - value.size(), ' ');
        card += value;
    }
    if (!comment.empty() && card.size() < 77) {
        card += " / ";
        card += comment.substr(0, kCard - card.size());
    }
    card.resize(kCard, ' ');
    return card;
}

inline string integer_card(const string& name, long long v, const string& comment) { return value_card(name, to_string(v), comment); }
inline string logical_card(const string& name, bool v, const string& comment) { return value_card(name, v ? "T" : "F", comment); }
inline string string_card(const string& name, const string& v, const string& comment) { return value_card(name, quoted(v), comment); }
inline string real_card(const string& name, double v, const string& comment) { return value_card(name, real_text(v), comment); }

inline void comment_cards(vector<string>& cards, const string& text)
{
    size_t at = 0;
    do {
        string card = "COMMENT " + text.substr(at, 72);
        card.resize(kCard, ' ');
        cards.push_back(card);
        at += 72;
    } while (at < text.size());
}

}   // namespace fits
}   // namespace krhost

template <typename T>
class FITSOutput {
public:
    explicit FITSOutput(const char* filename, bool clobber = true) { begin(filename, clobber); }
    explicit FITSOutput(const string& filename, bool clobber = true) { begin(filename.c_str(), clobber); }
    ~FITSOutput() { finish(false); }
    FITSOutput(const FITSOutput&) = delete;
    FITSOutput& operator=(const FITSOutput&) = delete;

    // Throws when any write of the file failed (the background writer records ENOSPC, EIO, ... and they surface here): a program that
    // calls close() -- imageplane_disc_image.cpp:340 does -- must not go on to report success over a truncated file.  (The destructor
    // only prints: it cannot throw.)
    void close() { finish(true); }

private:
    void finish(bool may_throw)
    {
        if (!open_) return;
        flush_hdu();
        for (int b = 0; b < 2; ++b) wait_for(b);
        for (int b = 0; b < 2; ++b) { free(buf_[b]); buf_[b] = nullptr; cap_[b] = 0; }
        if (::close(fd_) != 0 && !io_error_) io_error_ = errno;
        fd_ = -1;
        open_ = false;
        if (!io_error_) return;
        const string what = string("writing the file failed: ") + strerror(io_error_);
        if (may_throw) throw FITSOutputException(what);
        cerr << "FITSOutput ERROR : " << what << endl;
    }

public:
    // an empty primary array, so that keywords can be attached to the file as a whole
    void create_primary()
    {
        need_open();
        cout << "Creating empty primary extension in FITS file" << endl;
        new_hdu(8, nullptr, 0);
    }

    // ------ images ------
    void write_image_array(double* frame, int Nx, int Ny)
    {
        uint64_t* out = begin_image(Nx, Ny);
        const size_t n = static_cast<size_t>(Nx) * Ny;
        team(n, 1 << 16, [=](size_t a, size_t b) {
            for (size_t i = a; i < b; ++i) out[i] = big_endian(frame[i]);
        });
    }

    // data[x][y] -> image with X along FITS axis 1 (left to right) and Y along axis 2 (bottom to top); `transpose` swaps the
    // axes; flip_x / flip_y mirror the SOURCE array's axes
    void write_image(T** data, int Nx, int Ny, bool transpose = false, bool flip_x = false, bool flip_y = false)
    {
        if (transpose) {
            uint64_t* img = begin_image(Ny, Nx);
            team(static_cast<size_t>(Nx), 8, [=](size_t a, size_t b) {
                for (size_t j = a; j < b; ++j) {
                    const T* row = data[flip_x ? Nx - 1 - static_cast<int>(j) : static_cast<int>(j)];
                    uint64_t* out = img + j * Ny;
                    for (int k = 0; k < Ny; ++k) out[k] = big_endian(static_cast<double>(row[flip_y ? Ny - 1 - k : k]));
                }
            });
            return;
        }
        // out[k * Nx + j] = data[x(j)][y(k)] is a transposition: walk it in 32 x 32 tiles so that both sides stay in cache; a thread owns
        // whole tile ROWS of the output (k0 bands), so no two threads share a cache line of it
        uint64_t* img = begin_image(Nx, Ny);
        constexpr int kTile = 32;
        const size_t bands = (static_cast<size_t>(Ny) + kTile - 1) / kTile;
        team(bands, 1, [=](size_t a, size_t b) {
            for (size_t band = a; band < b; ++band) {
                const int k0 = static_cast<int>(band) * kTile, k1 = k0 + kTile < Ny ? k0 + kTile : Ny;
                for (int j0 = 0; j0 < Nx; j0 += kTile) {
                    const int j1 = j0 + kTile < Nx ? j0 + kTile : Nx;
                    for (int j = j0; j < j1; ++j) {
                        const T* row = data[flip_x ? Nx - 1 - j : j];
                        for (int k = k0; k < k1; ++k)
                            img[static_cast<size_t>(k) * Nx + j] = big_endian(static_cast<double>(row[flip_y ? Ny - 1 - k : k]));
                    }
                }
            }
        });
    }

    // one image extension per frame of data[frame][y][x], both axes mirrored (reference :190-207)
    void write_me_data_cube(T*** data, int Nframes, int Nx, int Ny)
    {
        cout << "Writing Frame..." << endl;
        vector<double> frame(static_cast<size_t>(Nx) * Ny);
        for (int f = 0; f < Nframes; ++f) {
            for (int j = 0; j < Ny; ++j)
                for (int k = 0; k < Nx; ++k) frame[static_cast<size_t>(Ny - 1 - j) * Nx + (Nx - 1 - k)] = data[f][j][k];
            write_image_array(frame.data(), Nx, Ny);
        }
    }

    // ------ tables: not on this path ------
    void create_table(const char*, int, long, char*[], char*[], char*[]) { throw FITSOutputException("binary tables are not implemented in this writer"); }
    void write_table_column(double*, long = -1, int = -1, long = 1, long = 1) { throw FITSOutputException("binary tables are not implemented in this writer"); }
    void write_table_column(int*, long = -1, int = -1, long = 1, long = 1) { throw FITSOutputException("binary tables are not implemented in this writer"); }

    // ------ keywords of the HDU created last ------
    void write_keyword(const char* keyname, const char* comment, int value) { add(krhost::fits::integer_card(keyname, value, comment)); }
    void write_keyword(const char* keyname, const char* comment, long value) { add(krhost::fits::integer_card(keyname, value, comment)); }
    void write_keyword(const char* keyname, const char* comment, double value) { add(krhost::fits::real_card(keyname, value, comment)); }
    void write_keyword(const char* keyname, const char* comment, const char* value) { add(krhost::fits::string_card(keyname, value, comment)); }
    void write_keyword(const char* keyname, const char* comment, char* value) { add(krhost::fits::string_card(keyname, value, comment)); }
    void write_keyword(const char* keyname, const char* comment, string value) { add(krhost::fits::string_card(keyname, value, comment)); }
    void write_keyword(const char* keyname, const char* comment, bool value) { add(krhost::fits::logical_card(keyname, value, comment)); }

    void write_comment(const char* comment)
    {
        need_hdu();
        krhost::fits::comment_cards(cards_, comment);
    }
    void write_comment(char* comment) { write_comment(static_cast<const char*>(comment)); }

    void set_ext_name(const char* extname) { add(krhost::fits::string_card("EXTNAME", extname, "Name of this extension")); }
    void set_ext_name(char* extname) { set_ext_name(static_cast<const char*>(extname)); }

private:
    void begin(const char* filename, bool clobber)
    {
        cout << "Opening FITS file for output: " << filename;
        if (clobber) cout << " (will overwrite if file exists)" << endl;
        if (clobber) remove(filename);
        else if (ifstream(filename).good()) throw FITSOutputException("Could not open file", 105);
        fd_ = ::open(filename, O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (fd_ < 0) throw FITSOutputException("Could not open file", 104);
        open_ = true;
    }
    void need_open() const
    {
        if (!open_) throw FITSOutputException("File is not open");
    }
    void need_hdu() const
    {
        need_open();
        if (hdus_ == 0) throw FITSOutputException("No extension has been created yet");
    }
    void add(const string& card)
    {
        need_hdu();
        cards_.push_back(card);
    }

    static uint64_t big_endian(double v)   // IEEE big-endian image of v as a word to be stored by a little-endian host
    {
        uint64_t bits;
        memcpy(&bits, &v, 8);
        return __builtin_bswap64(bits);
    }

    // a BITPIX = -64 image HDU of Nx x Ny pixels; the caller fills EVERY word of the returned buffer (it is not zeroed)
    uint64_t* begin_image(int Nx, int Ny)
    {
        need_open();
        cout << "Adding " << Nx << 'x' << Ny << " image extension to FITS file" << endl;
        const long axes[2] = {Nx, Ny};
        new_hdu(-64, axes, 2);
        cur_ ^= 1;
        wait_for(cur_);                              // the writer that last used this buffer
        words_ = static_cast<size_t>(Nx) * Ny;
        const size_t padded = (words_ * 8 + krhost::fits::kBlock - 1) / krhost::fits::kBlock * krhost::fits::kBlock;
        if (cap_[cur_] < padded) {
            free(buf_[cur_]);
            buf_[cur_] = static_cast<uint64_t*>(malloc(padded));
            if (!buf_[cur_]) throw FITSOutputException("out of memory for an image buffer");
            cap_[cur_] = padded;
        }
        memset(reinterpret_cast<char*>(buf_[cur_]) + words_ * 8, 0, padded - words_ * 8);      // the block padding goes out with the data
        have_data_ = true;
        return buf_[cur_];
    }

    static int team_size()
    {
        unsigned hw = thread::hardware_concurrency();
        int t = hw == 0 ? 4 : static_cast<int>(hw < 16 ? hw : 16);
        if (const char* e = getenv("KRTRACE_HOST_THREADS")) {
            const int v = atoi(e);
            if (v >= 1 && v < t) t = v;
        }
        return t;
    }

    // fn(a, b) over [0, n) in contiguous pieces, one per thread; the calling thread takes the first piece
    template <class F>
    static void team(size_t n, size_t min_per_thread, F fn)
    {
        size_t t = static_cast<size_t>(team_size());
        if (min_per_thread > 0 && n / min_per_thread < t) t = n / min_per_thread;
        if (t <= 1) { fn(static_cast<size_t>(0), n); return; }
        vector<thread> others;
        others.reserve(t - 1);
        for (size_t i = 1; i < t; ++i) others.emplace_back(fn, n * i / t, n * (i + 1) / t);
        fn(static_cast<size_t>(0), n / t);
        for (thread& th : others) th.join();
    }

    static int write_all(int fd, const char* p, size_t bytes, off_t at)
    {
        while (bytes > 0) {
            const ssize_t w = ::pwrite(fd, p, bytes, at);
            if (w < 0) {
                if (errno == EINTR) continue;
                return errno;
            }
            p += w; bytes -= static_cast<size_t>(w); at += w;
        }
        return 0;
    }

    void wait_for(int b)
    {
        if (writer_[b].joinable()) writer_[b].join();
        if (writer_error_[b] && !io_error_) io_error_ = writer_error_[b];
        writer_error_[b] = 0;
    }

    // mandatory cards of a new image HDU; the previous one goes to disk first
    void new_hdu(int bitpix, const long* axes, int naxis)
    {
        namespace kf = krhost::fits;
        flush_hdu();
        if (hdus_ == 0) {
            cards_.push_back(kf::logical_card("SIMPLE", true, "file does conform to FITS standard"));
        } else {
            cards_.push_back(kf::string_card("XTENSION", "IMAGE", "IMAGE extension"));
        }
        cards_.push_back(kf::integer_card("BITPIX", bitpix, "number of bits per data pixel"));
        cards_.push_back(kf::integer_card("NAXIS", naxis, "number of data axes"));
        for (int i = 0; i < naxis; ++i)
            cards_.push_back(kf::integer_card("NAXIS" + to_string(i + 1), axes[i], "length of data axis " + to_string(i + 1)));
        if (hdus_ == 0) {
            cards_.push_back(kf::logical_card("EXTEND", true, "FITS dataset may contain extensions"));
            kf::comment_cards(cards_, "  FITS (Flexible Image Transport System) format is defined in 'Astronomy");
            kf::comment_cards(cards_, "  and Astrophysics', volume 376, page 359; bibcode: 2001A&A...376..359H");
        } else {
            cards_.push_back(kf::integer_card("PCOUNT", 0, "required keyword; must = 0"));
            cards_.push_back(kf::integer_card("GCOUNT", 1, "required keyword; must = 1"));
        }
        ++hdus_;
    }

    void flush_hdu()
    {
        namespace kf = krhost::fits;
        if (cards_.empty()) return;
        string end = "END";
        end.resize(kf::kCard, ' ');
        cards_.push_back(end);
        string header;
        header.reserve((cards_.size() + 36) * kf::kCard);
        for (string& c : cards_) {
            for (char& ch : c)                       // header bytes are printable ASCII; anything else becomes a blank
                if (static_cast<unsigned char>(ch) < 32 || static_cast<unsigned char>(ch) > 126) ch = ' ';
            header.append(c.data(), kf::kCard);
        }
        const size_t used = header.size() % kf::kBlock;
        if (used) header.append(kf::kBlock - used, ' ');
        if (const int e = write_all(fd_, header.data(), header.size(), pos_)) {
            if (!io_error_) io_error_ = e;
        }
        pos_ += static_cast<off_t>(header.size());
        if (have_data_) {
            // the image leaves in the background, in parallel slices, while the application builds the next one in the other buffer
            const size_t bytes = (words_ * 8 + kf::kBlock - 1) / kf::kBlock * kf::kBlock;
            const char* src = reinterpret_cast<const char*>(buf_[cur_]);
            const int fd = fd_, b = cur_;
            const off_t at = pos_;
            int* err = &writer_error_[b];
            writer_[b] = thread([=]() {
                constexpr size_t kSlice = size_t(8) << 20;
                const size_t slices = (bytes + kSlice - 1) / kSlice;
                const int writers = slices < 4 ? static_cast<int>(slices) : 4;
                vector<int> errs(static_cast<size_t>(writers), 0);
                vector<thread> ths;
                for (int w = 0; w < writers; ++w)
                    ths.emplace_back([=, &errs]() {
                        for (size_t s = static_cast<size_t>(w); s < slices && !errs[static_cast<size_t>(w)]; s += static_cast<size_t>(writers)) {
                            const size_t off = s * kSlice, len = off + kSlice < bytes ? kSlice : bytes - off;
                            errs[static_cast<size_t>(w)] = write_all(fd, src + off, len, at + static_cast<off_t>(off));
                        }
                    });
                for (thread& t : ths) t.join();
                for (int e : errs)
                    if (e && !*err) *err = e;
            });
            pos_ += static_cast<off_t>(bytes);
            have_data_ = false;
        }
        cards_.clear();
    }

    int fd_ = -1;
    off_t pos_ = 0;              // where the next HDU starts
    bool open_ = false;
    int hdus_ = 0;
    int io_error_ = 0;
    vector<string> cards_;
    uint64_t* buf_[2] = {nullptr, nullptr};   // big-endian words of the current image / of the one being written
    size_t cap_[2] = {0, 0};
    size_t words_ = 0;
    int cur_ = 0;
    bool have_data_ = false;
    thread writer_[2];
    int writer_error_[2] = {0, 0};
};

#endif /* FITS_OUTPUT_H_ */
