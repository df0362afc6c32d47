// include/par_file.h -- `key = value` parameter files, API of the reference's src/include/par_file.h:37-205.
//
// File format (reference :43-122): one assignment per line, `#` starts a comment, blank lines are skipped, keys and
// values are stripped of blanks and tabs; a repeated key keeps its FIRST value and a note goes to stderr.
// Getters: see key_value_store.h.  Part of the drop-in tree: the reference's applications include
// this file as "include/par_file.h" / "../include/par_file.h".
#ifndef RAYTRACE_PAR_FILE_H_H
#define RAYTRACE_PAR_FILE_H_H

#include <fstream>
#include <iostream>

#include "key_value_store.h"
using namespace std;   // the reference's utility headers do this and its applications rely on it

class ParameterException : public krhost::OptionError {
public:
    explicit ParameterException(const string& msg) : krhost::OptionError("ParameterFile ERROR : " + msg) {}
};

class ParameterFile : public krhost::KeyValueStore<ParameterException> {
public:
    explicit ParameterFile(const string& filename)
    {
        ifstream in(filename.c_str());
        if (!in.is_open()) throw ParameterException("ParameterFile ERROR: Could not open file " + filename);
        string line, key, value;
        while (getline(in, line)) {
            const size_t hash = line.find('#');
            if (hash != string::npos) line.erase(hash);
            if (line.find_first_not_of(' ') == string::npos) continue;
            if (!krhost::split_assignment(line, key, value)) {
                // the reference stores such a line under the whole line as key (:86-97); nothing can ask for it
                key = krhost::strip_blanks(line);
                value = key;
            }
            if (!store(key, value)) cerr << "ParameterFile ERROR: Duplicate definition of " << key << endl;
        }
    }

    // the stored text of a key (the reference returns a pointer into a temporary, :182-189; this one stays valid)
    char* get_string_parameter(const string& key) const { return const_cast<char*>(lookup(key).c_str()); }

protected:
    string missing_text(const string& key) const override { return "ParameterFile ERROR: " + key + " not found in parameter file"; }
};

#endif /* RAYTRACE_PAR_FILE_H_H */
