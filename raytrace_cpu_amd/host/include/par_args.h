// include/par_args.h -- `--key=value` command-line options, API of the reference's src/include/par_args.h:38-230.
//
// An argument containing "--" is an option and is stored under its text before '=' (prefix included, e.g.
// "--outfile"); a repeated option throws; every other argument is positional (operator[]).  Getters: key_value_store.h.
#ifndef RAYTRACE_PAR_ARGS_H_H
#define RAYTRACE_PAR_ARGS_H_H

#include <iostream>
#include <vector>

#include "key_value_store.h"
using namespace std;

class ArgumentException : public krhost::OptionError {
public:
    explicit ArgumentException(const string& msg) : krhost::OptionError("Argument ERROR : " + msg) {}
};

class ParameterArgs : public krhost::KeyValueStore<ArgumentException> {
public:
    ParameterArgs(int argc, char** argv)
    {
        for (int i = 1; i < argc; ++i) {
            const string arg(argv[i]);
            if (arg.find("--") == string::npos) {
                positional_.push_back(arg);
                continue;
            }
            string key, value;
            if (!krhost::split_assignment(arg, key, value)) {   // a bare "--flag": present, value = its own text (:99-110)
                key = krhost::strip_blanks(arg);
                value = key;
            }
            if (!store(key, value)) throw ArgumentException("Duplicate definition of " + key);
        }
    }

    string get_string_parameter(const string& key) const { return lookup(key); }

    int num_positional() const { return static_cast<int>(positional_.size()); }

    string operator[](int i) const
    {
        if (i < 0 || i >= num_positional()) {
            ostringstream msg;
            msg << "Positional argument " << i + 1 << " not supplied";
            throw ArgumentException(msg.str());
        }
        return positional_[i];
    }

protected:
    string missing_text(const string& key) const override { return "Required parameter " + key + " not specified"; }

private:
    vector<string> positional_;
};

#endif /* RAYTRACE_PAR_ARGS_H_H */
