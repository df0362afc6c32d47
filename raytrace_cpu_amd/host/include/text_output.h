// include/text_output.h -- columnar text writer, API and byte format of the reference's src/include/text_output.h:19-127.
//
// Format (what makes an output file byte-comparable): every value goes through `operator<<` of a std::ofstream with
// `scientific` float format, precision 8 and field width 20, right-aligned, no separator; a row ends with `endl`.
// So a double prints as "      1.23697066e+00", an integer as its decimal digits, NaN as "nan" / "-nan" with the
// sign bit the value carries.  Used by the reference's emissivity applications for their 7-column .dat tables
// (emissivity.cpp:136-147).
#ifndef TEXT_OUTPUT_H_
#define TEXT_OUTPUT_H_

#include <fstream>
#include <iomanip>
#include <iostream>
#include <string>
using namespace std;

class TextOutput {
public:
    TextOutput(const char* filename, bool append = false, int precision = 8, int colwidth = 20, ios_base::fmtflags format = ios::scientific)
        : width_(colwidth)
    {
        start(filename, append, precision, format);
    }
    TextOutput(const string& filename, bool append = false, int precision = 8, int colwidth = 20, ios_base::fmtflags format = ios::scientific)
        : width_(colwidth)
    {
        start(filename.c_str(), append, precision, format);
    }
    ~TextOutput() { close(); }

    void close()
    {
        if (file_.is_open()) file_.close();
        usable_ = false;
    }

    void set_format(ios_base::fmtflags format = ios::scientific) { file_.setf(format); }
    void set_precision(int precision = 8) { file_.precision(precision); }

    void newline(int n = 1)
    {
        while (n-- > 0) file_ << endl;
    }

    // manipulators (endl)
    TextOutput& operator<<(ostream& (*manip)(ostream&))
    {
        if (usable_) file_ << manip;
        else complain();
        return *this;
    }

    // one column
    template <typename V>
    TextOutput& operator<<(const V& value)
    {
        if (usable_) file_ << setw(width_) << value;
        else complain();
        return *this;
    }

private:
    void start(const char* filename, bool append, int precision, ios_base::fmtflags format)
    {
        cout << "Writing results to text file: " << filename << endl;
        file_.open(filename, append ? ios::app : ios::out);
        usable_ = static_cast<bool>(file_);
        if (!usable_) {
            cerr << "************" << endl << "TextOutput ERROR: Could not open output file " << filename << endl << "************" << endl;
            return;
        }
        file_.setf(format);
        file_.precision(precision);
    }
    static void complain() { cerr << "TextOutput ERROR: File is not open" << endl; }

    ofstream file_;
    bool usable_ = false;
    int width_;
};

#endif /* TEXT_OUTPUT_H_ */
